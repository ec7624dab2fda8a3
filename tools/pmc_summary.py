#!/usr/bin/env python
"""Mean value per launch of every counter found under the given rocprofv3 --pmc output directories, for the kernels
whose name contains one of the given substrings.

    tools/pmc_summary.py <dir> [<dir> ...] -- <kernel substring> [...]
"""
import csv
import glob
import json
import sys

args = sys.argv[1:]
dirs, kernels = args[:args.index("--")], args[args.index("--") + 1:]
out = {}
for d in dirs:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            for k in kernels:
                if k in r["Kernel_Name"]:
                    out.setdefault(k, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
res = {k: {c: {"mean_per_launch": sum(v) / len(v), "launches": len(v)} for c, v in cs.items()} for k, cs in out.items()}
print(json.dumps(res, indent=1))
