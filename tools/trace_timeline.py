#!/usr/bin/env python3
"""Timeline of a rocprofv3 --kernel-trace CSV: one line per kernel dispatch with queue, start and end
(µs, relative to the first dispatch shown) and the kernels it overlapped with.
    python tools/trace_timeline.py <..._kernel_trace.csv> [first_dispatch [count]]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
count = int(sys.argv[3]) if len(sys.argv) > 3 else 40
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if "va::" in r["Kernel_Name"] or "ccl" in r["Kernel_Name"]]
win = rows[first:first + count]
t0 = int(win[0]["Start_Timestamp"])
def short(nm):
    nm = nm.split("(")[0].replace("void ", "").replace("va::", "").replace("(anonymous namespace)::", "")
    return nm[:34]
for i, r in enumerate(win):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    over = [short(q["Kernel_Name"]) + " %.0fus" % ((min(e, int(q["End_Timestamp"])) - max(s, int(q["Start_Timestamp"]))) / 1e3)
            for q in win if q is not r and int(q["Start_Timestamp"]) < e and int(q["End_Timestamp"]) > s]
    print("%-36s q%-3s %9.1f -> %9.1f  (%7.1f us)  %s" % (short(r["Kernel_Name"]), r.get("Queue_Id", "?"), (s - t0) / 1e3,
                                                        (e - t0) / 1e3, (e - s) / 1e3, "|| " + ", ".join(over) if over else ""))
