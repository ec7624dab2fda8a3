#!/usr/bin/env python
"""Summarise `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes of bench.py into
profiles/pmc_traffic.json: HBM bytes per launch of the dominant kernel, corrected as
/opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes for gfx950: FETCH_SIZE reports
half of a wide coalesced read stream -> x2; WRITE_SIZE is exact; both are in KB.

    tools/pmc_traffic.py <fetch_dir> <write_dir> <kernel substring> <out.json> <kernel source file in ips_amd/csrc> <patches per launch>

The record carries a hash of the kernel's source file; bench.py reports `traffic` only while that file is unchanged.
"""
import csv
import glob
import hashlib
import json
import os
import sys


def mean_counter(d, name, kernel):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f))
         if r["Counter_Name"] == name and kernel in r["Kernel_Name"]]
    return sum(v) / len(v), len(v)


fetch_dir, write_dir, kernel, out, source, per_launch = sys.argv[1:7]
src_path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ips_amd", "csrc", source)
fetch_kb, n1 = mean_counter(fetch_dir, "FETCH_SIZE", kernel)
write_kb, n2 = mean_counter(write_dir, "WRITE_SIZE", kernel)
res = {"kernel": kernel, "launches_averaged": [n1, n2],
       "FETCH_SIZE_KB_raw": fetch_kb, "WRITE_SIZE_KB": write_kb,
       "hbm_read_bytes": 2 * fetch_kb * 1024, "hbm_write_bytes": write_kb * 1024,
       "hbm_bytes_per_launch": 2 * fetch_kb * 1024 + write_kb * 1024,
       "correction": "FETCH_SIZE x2 (gfx950 reports half of a wide coalesced read), WRITE_SIZE x1, KB = 1024 B",
       "source": source, "source_sha16": hashlib.sha256(open(src_path, "rb").read()).hexdigest()[:16],
       "patches_per_launch": int(per_launch),
       "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE -- python3 bench.py --steps 3 --warmup 1 --cpu-seconds 0"}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res))
