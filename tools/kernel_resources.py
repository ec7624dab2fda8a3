#!/usr/bin/env python
"""Registers, spills, scratch and LDS of every kernel of libipsx, from the compiler's own metadata (hipcc -S of each
translation unit with the Makefile's flags): the table VERDICT r03 asks for under profiles/.
    python tools/kernel_resources.py > profiles/r04_kernel_resources.txt"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ips_amd", "csrc")
FLAGS = "--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -ffp-contract=off -fno-fast-math --cuda-device-only -S".split()
rows = []
for src in sorted(f for f in os.listdir(CSRC) if f.endswith(".hip")):
    with tempfile.NamedTemporaryFile(suffix=".s") as tmp:
        subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + [os.path.join(CSRC, src), "-o", tmp.name], check=True,
                       stderr=subprocess.DEVNULL)
        text = open(tmp.name).read()
    for m in re.finditer(r"- \.agpr_count:\s+(\d+).*?\.group_segment_fixed_size:\s+(\d+).*?\.name:\s+(\S+).*?\.private_segment_fixed_size:\s+(\d+)"
                         r".*?\.sgpr_count:\s+(\d+).*?\.sgpr_spill_count:\s+(\d+).*?\.vgpr_count:\s+(\d+).*?\.vgpr_spill_count:\s+(\d+)", text, re.S):
        agpr, lds, name, scratch, sgpr, sspill, vgpr, vspill = m.groups()
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        dem = re.sub(r"\(.*", "", dem).replace("void ", "").replace("ipsx::", "")
        rows.append((src, dem, int(vgpr), int(agpr), int(vspill), int(scratch), int(sgpr), int(sspill), int(lds)))
print("%-18s %-64s %5s %5s %7s %8s %5s %7s %7s" % ("file", "kernel", "vgpr", "agpr", "v-spill", "scratchB", "sgpr", "s-spill", "ldsB"))
for r in rows:
    print("%-18s %-64s %5d %5d %7d %8d %5d %7d %7d" % ((r[0], r[1][:64]) + r[2:]))
