#!/usr/bin/env python3
"""Copies the judged summaries of one tools/profile_round.sh run from gpurun_out/<tag>/ into profiles/:
    python tools/collect_profile.py <tag> <workload>
-> profiles/<tag>_kernel_stats.csv (rocprofv3 --kernel-trace --stats), <tag>_bench.json (unprofiled bench
line of the same command), <tag>_pmc_summary.csv + profiles/traffic.json (FETCH_SIZE / WRITE_SIZE passes)"""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, workload = sys.argv[1], sys.argv[2]
src = os.path.join(ROOT, "gpurun_out", tag)
shutil.copy(os.path.join(src, "trace", "t_kernel_stats.csv"), os.path.join(ROOT, "profiles", tag + "_kernel_stats.csv"))
shutil.copy(os.path.join(src, "bench.json"), os.path.join(ROOT, "profiles", tag + "_bench.json"))
subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "pmc_to_traffic.py"), os.path.join(src, "pmc"),
                       os.path.join(ROOT, "profiles", tag), workload])
