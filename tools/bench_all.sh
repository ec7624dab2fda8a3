#!/bin/bash
# Every bench line of DESIGN.md section 6 in one go (one MI355X, ~5 minutes): the headline, the other BASELINE workloads (each
# with its own cpu_baseline on a bounded sample) and the secondary modes of the headline workload, one JSON line each
# (stdout), each with its parity object.
#   bash tools/bench_all.sh > profiles/rNN_bench_all.jsonl
set -e
cd "$(dirname "$0")/.."
python bench.py 2>/dev/null
for c in b1 mnist3000 native50 traffic cam cam_native; do python bench.py --config $c --cpu-seconds 8 2>/dev/null; done
python bench.py --config cam --batch 8 --cpu-seconds 0 2>/dev/null      # 8 / 16 slides per call: the projector's rate (the loops hide)
python bench.py --config cam --batch 16 --cpu-seconds 0 2>/dev/null
IPSX_IMAGE_STREAM=0 python bench.py --config b1 --cpu-seconds 0 2>/dev/null                     # one image in parts ...
IPSX_IMAGE_STREAM=0 python bench.py --config b1 --cpu-seconds 0 --no-kernel-events 2>/dev/null  # ... and without the bench's own event records
python bench.py --precision fp32x3 --cpu-seconds 0 2>/dev/null
python bench.py --precision bf16 --cpu-seconds 0 2>/dev/null
python bench.py --precision bf16 --storage f16 --cpu-seconds 0 2>/dev/null      # BASELINE configs[4] as written: fp16-stored patches
python bench.py --lazy --cpu-seconds 0 2>/dev/null
python bench.py --dedup-blank --cpu-seconds 0 2>/dev/null
