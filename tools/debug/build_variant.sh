#!/bin/bash
# Ablation / experiment build of ONE kernel source into video-analysis_amd/csrc/build/dbg/lib_<name>.so
# (linked with the regular objects of every other source; run `make -C video-analysis_amd/csrc` first):
#   tools/debug/build_variant.sh va_gauss_f32_fused.hip skew2 -DROWIS_SKEW=2 -DROWIS_SKEW_UNITS=1
# then on the GPU box:  python tools/debug/bench_dbg.py lib_skew2.so --workload ...
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
C=$ROOT/video-analysis_amd/csrc
SRC=$1; NAME=$2; shift 2
mkdir -p $C/build/dbg
EXTRA=""
[ "$SRC" = "va_gauss_f32_fused.hip" ] && EXTRA="-fno-slp-vectorize"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math $EXTRA "$@" -c $C/$SRC -o $C/build/dbg/$NAME.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $C/build/dbg/lib_$NAME.so $(ls $C/build/*.o | grep -v "${SRC%.hip}.o") $C/build/dbg/$NAME.o -ldl
echo $C/build/dbg/lib_$NAME.so
