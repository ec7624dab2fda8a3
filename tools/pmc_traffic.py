#!/usr/bin/env python
"""Summarise `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes of bench.py into profiles/pmc_traffic.json (one
record per workload): HBM bytes, corrected as /opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes for
gfx950: FETCH_SIZE reports half of a wide coalesced read stream -> x2; WRITE_SIZE is exact; both are in KB.

    tools/pmc_traffic.py <fetch_dir> <write_dir> <workload> <ips() calls in the profiled run> <kernel sources, comma separated>
                         [<dominant kernel substring(s), '|' separated> <patches per launch>]

With a dominant kernel (the fused trunk: the encoder IS one kernel) the record holds bytes per launch of that kernel;
otherwise bytes per step summed over every kernel of the run (calls = 1 + warmup + 2 * steps of bench.py: the first
call, the warm-up, the timed loop and the synchronised-latency loop).  The record carries a hash of the kernel sources;
bench.py reports `traffic` only while they are unchanged.
"""
import csv
import glob
import hashlib
import json
import os
import sys
from collections import defaultdict

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def per_kernel(d, name):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    acc = defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == name:
            k = r["Kernel_Name"].split("(")[0][:80]
            acc[k][0] += float(r["Counter_Value"])
            acc[k][1] += 1
    return acc


def main():
    fetch_dir, write_dir, workload, calls, sources = sys.argv[1:6]
    calls = int(calls)
    dominant = sys.argv[6] if len(sys.argv) > 6 else None
    fetch, write = per_kernel(fetch_dir, "FETCH_SIZE"), per_kernel(write_dir, "WRITE_SIZE")
    h = hashlib.sha256()
    for src in sources.split(","):
        h.update(open(os.path.join(REPO, "ips_amd", "csrc", src), "rb").read())
    rec = {"correction": "FETCH_SIZE x2 (gfx950 reports half of a wide coalesced read), WRITE_SIZE x1, KB = 1024 B",
           "source": sources, "source_sha16": h.hexdigest()[:16],
           "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE -- python3 bench.py --config %s --steps 3 --warmup 1 --cpu-seconds 0" % workload}
    if dominant:
        names = dominant.split("|")                # e.g. the fused trunk's two exact kernels (whole rounds | remainder)
        fk = [v for k, v in fetch.items() if any(n in k for n in names)]
        wk = [v for k, v in write.items() if any(n in k for n in names)]
        fetch_kb = sum(v[0] for v in fk) / sum(v[1] for v in fk)
        write_kb = sum(v[0] for v in wk) / sum(v[1] for v in wk)
        rec.update({"kernel": dominant, "launches_averaged": [sum(v[1] for v in fk), sum(v[1] for v in wk)],
                    "FETCH_SIZE_KB_raw": fetch_kb, "WRITE_SIZE_KB": write_kb,
                    "hbm_read_bytes": 2 * fetch_kb * 1024, "hbm_write_bytes": write_kb * 1024,
                    "hbm_bytes_per_launch": 2 * fetch_kb * 1024 + write_kb * 1024,
                    "patches_per_launch": int(sys.argv[7])})
    else:
        rd = sum(v[0] for v in fetch.values()) * 2 * 1024 / calls
        wr = sum(v[0] for v in write.values()) * 1024 / calls
        top = sorted(set(fetch) | set(write), key=lambda k: -(2 * fetch[k][0] + write[k][0]))[:8]
        rec.update({"ips_calls": calls, "hbm_read_bytes_per_step": rd, "hbm_write_bytes_per_step": wr,
                    "hbm_bytes_per_step": rd + wr,
                    "kernels": [{"kernel": k, "launches_per_step": fetch[k][1] / calls,
                                 "read_bytes_per_step": 2 * 1024 * fetch[k][0] / calls,
                                 "write_bytes_per_step": 1024 * write[k][0] / calls} for k in top]})
    out = os.path.join(REPO, "profiles", "pmc_traffic.json")
    allrec = json.load(open(out)) if os.path.exists(out) else {}
    if "kernel" in allrec:                     # the single-record file of rounds 1-2: it was the headline's
        allrec = {"mnist": allrec}
    allrec[workload] = rec
    json.dump(allrec, open(out, "w"), indent=1)
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
