#!/bin/bash
# SQ counter passes of the headline chain's kernels (instruction mix, MFMA busy, LDS, waits)
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_gauss
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC"; do
  tag=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --pmc $grp --output-format csv -d $OUT -o $tag -- python3 $ROOT/bench.py --workload cfg3_1080p_full_chain --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/$tag.err || { tail -5 $OUT/$tag.err; }
done
python3 - <<PY
import csv, glob, collections, re
for f in sorted(glob.glob("$OUT/*counter_collection.csv")):
    acc = collections.defaultdict(lambda: [0, 0])
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "va::" not in k: continue
        m = re.search(r"(\w+_kernel)", k); k = m.group(1) if m else k[:30]
        a = acc[(k, r["Counter_Name"])]; a[0] += float(r["Counter_Value"]); a[1] += 1
    for (k, c), (v, n) in sorted(acc.items()):
        print("%-26s %-26s %16.0f per launch" % (k, c, v / n))
PY
