#!/usr/bin/env python3
"""Copies the judged summaries of one tools/profile_round.sh run from gpurun_out/<tag>/ into profiles/:
    python tools/collect_profile.py <tag> <workload>
-> profiles/<tag>_kernel_stats.csv (rocprofv3 --kernel-trace --stats), <tag>_bench.json (unprofiled bench
line of the same command), <tag>_pmc_summary.csv + profiles/traffic.json (FETCH_SIZE / WRITE_SIZE passes)"""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, workload = sys.argv[1], sys.argv[2]
src = os.path.join(ROOT, "gpurun_out", tag)
shutil.copy(os.path.join(src, "trace", "t_kernel_stats.csv"), os.path.join(ROOT, "profiles", tag + "_kernel_stats.csv"))
shutil.copy(os.path.join(src, "bench.json"), os.path.join(ROOT, "profiles", tag + "_bench.json"))
shutil.copy(os.path.join(src, tag + "_pmc_summary.csv"), os.path.join(ROOT, "profiles", tag + "_pmc_summary.csv"))
new = json.load(open(os.path.join(src, "traffic.json")))
tpath = os.path.join(ROOT, "profiles", "traffic.json")
allt = json.load(open(tpath)) if os.path.exists(tpath) else {}
allt[workload] = new[workload]
allt.setdefault("_source", {})[workload] = tag
json.dump(allt, open(tpath, "w"), indent=1, sort_keys=True)
print(json.dumps(allt[workload], indent=1))
