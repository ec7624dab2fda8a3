#!/usr/bin/env python
"""Print the kernel timeline (start, duration, name) of the LAST `n` kernel dispatches of a rocprofv3 --kernel-trace run.
    tools/timeline.py <dir> [n]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows = rows[-n:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.1f us  +%8.1f us  q%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?"), r["Kernel_Name"][:70]))
