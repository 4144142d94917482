#!/bin/bash
# rocprofv3 evidence for one bench.py workload (run on the GPU box, from the repo root):
#   tools/profile_round.sh <workload> <tag>        e.g.  tools/profile_round.sh cfg3_1080p_full_chain r02_a
# 1. kernel trace + stats of `python3 bench.py --workload W`, 2. an unprofiled bench line,
# 3. two --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs) -- all into gpurun_out/<tag>/
set -o pipefail
W=$1; TAG=$2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT $ROOT/profiles
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 $ROOT/bench.py --workload $W --steps 10 --warmup 3 --no-cpu-baseline --no-check --no-extra > $OUT/bench_traced.json 2> $OUT/trace.err || { tail -5 $OUT/trace.err; exit 1; }
python3 $ROOT/bench.py --workload $W --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err || { tail -5 $OUT/bench.err; exit 1; }
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc -o fetch -- python3 $ROOT/bench.py --workload $W --steps 3 --warmup 1 --no-cpu-baseline --no-check --no-extra > /dev/null 2> $OUT/pmc_fetch.err || { tail -5 $OUT/pmc_fetch.err; exit 1; }
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc -o write -- python3 $ROOT/bench.py --workload $W --steps 3 --warmup 1 --no-cpu-baseline --no-check --no-extra > /dev/null 2> $OUT/pmc_write.err || { tail -5 $OUT/pmc_write.err; exit 1; }
# per-kernel HBM bytes are summarised here (the raw per-dispatch rows are tens of MB; gpurun brings back 64 MiB)
python3 $ROOT/tools/pmc_to_traffic.py $OUT/pmc $OUT/$TAG $W > /dev/null && rm -f $OUT/pmc/*counter_collection.csv $OUT/pmc/*agent_info.csv
# the start/end timestamps of the traced run stay as they are (one row per dispatch, a few hundred KB)
# everything stays under gpurun_out/<tag>/ (the only directory gpurun brings back); afterwards, in the
# build container:  python tools/collect_profile.py <tag> <workload>   -> profiles/<tag>_*.csv|json, traffic.json
grep "va::" $OUT/trace/t_kernel_stats.csv | cut -d, -f1-4 | cut -c1-150; cut -c1-300 $OUT/bench.json
