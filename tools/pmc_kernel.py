#!/usr/bin/env python3
"""Per-kernel averages of a rocprofv3 --pmc counter_collection.csv (values / 1e6).
    python tools/pmc_kernel.py <counter_collection.csv> [kernel-substring ...]"""
import collections
import csv
import re
import sys


def main():
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(sys.argv[1])):
        m = re.search(r"va::\(anonymous namespace\)::(\w+)", r["Kernel_Name"])
        if m:
            agg[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
    want = sys.argv[2:]
    for k in sorted(agg):
        if want and not any(w in k for w in want):
            continue
        print(k, {c: round(sum(v) / len(v) / 1e6, 3) for c, v in sorted(agg[k].items())})


if __name__ == "__main__":
    main()
