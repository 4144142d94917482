#!/bin/bash
# SQ-counter evidence for one bench.py workload (run on the GPU box, from the repo root):
#   tools/profile_sq.sh <workload> <tag>       e.g.  tools/profile_sq.sh cfg3_1080p_full_chain r03_sq_cfg3
# Separate rocprofv3 --pmc passes (counters only: no trace domains), four counters per pass; the
# per-kernel averages go to gpurun_out/<tag>/<tag>_sq_summary.csv  (copy it into profiles/).
set -o pipefail
W=$1; TAG=$2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" \
           "SQ_INSTS_VALU_MFMA_I8 SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM" \
           "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD" \
           "SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_INSTS_GDS SQ_INSTS_BRANCH" \
           "GRBM_GUI_ACTIVE SQ_LEVEL_WAVES SQ_ACCUM_PREV_HIRES SQ_CYCLES"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $OUT -o p$i -- python3 $ROOT/bench.py --workload $W --steps 3 --warmup 1 --no-cpu-baseline --no-check --no-extra > /dev/null 2> $OUT/p$i.err || { echo "pass $i ($grp) failed:"; tail -3 $OUT/p$i.err; }
done
python3 - <<PY
import csv, glob, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("$OUT/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        m = re.search(r"va::\(anonymous namespace\)::(?:\w+::)*(\w+)", r["Kernel_Name"])
        if m:
            acc[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = sorted({c for k in acc for c in acc[k]})
with open("$OUT/${TAG}_sq_summary.csv", "w") as fh:
    fh.write("# rocprofv3 --pmc passes of: python3 bench.py --workload $W --steps 3 --warmup 1 (averages per launch, all XCDs summed)\n")
    fh.write("kernel,launches," + ",".join(names) + "\n")
    for k in sorted(acc):
        n = max(len(v) for v in acc[k].values())
        fh.write(k + "," + str(n) + "," + ",".join("%.0f" % (sum(acc[k][c]) / len(acc[k][c])) if c in acc[k] else "" for c in names) + "\n")
print(open("$OUT/${TAG}_sq_summary.csv").read())
PY
rm -f $OUT/*counter_collection.csv $OUT/*agent_info.csv      # (raw per-dispatch rows: tens of MB; gpurun brings back 64 MiB)
