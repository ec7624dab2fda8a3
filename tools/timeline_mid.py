#!/usr/bin/env python
"""Kernel timeline of dispatches [i0, i0 + n) counted from the END of a rocprofv3 --kernel-trace run (csv):
    tools/timeline_mid.py <dir> <back> <n>     e.g. 400 40 = forty dispatches starting 400 before the last"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
back, n = int(sys.argv[2]), int(sys.argv[3])
rows = rows[len(rows) - back:len(rows) - back + n]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.1f us  +%8.1f us  q%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?"), r["Kernel_Name"][:70]))
