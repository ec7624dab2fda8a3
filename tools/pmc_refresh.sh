#!/bin/bash
# The counter passes of tools/collect_profiles.sh for SOME workloads (after a change to their kernels' source files:
# bench.py reports `roofline.traffic` only while profiles/pmc_traffic.json carries the hash of the sources it runs):
#   bash tools/pmc_refresh.sh <out dir under gpurun_out> mnist traffic native50
set -u
cd "$(dirname "$0")/.."
OUT=${1:-gpurun_out/pmc}; shift
mkdir -p "$OUT"
export TMPDIR=/tmp
for c in "$@"; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_${c}_$ctr
    (cd /tmp && rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/pmc_${c}_$ctr -o pmc -- python3 "$OLDPWD/bench.py" --config $c --steps 3 --warmup 1 --cpu-seconds 0 > /dev/null 2>&1)
  done
  case $c in
    mnist) python tools/pmc_traffic.py /tmp/pmc_mnist_FETCH_SIZE /tmp/pmc_mnist_WRITE_SIZE mnist 8 fused_trunk.hip,fused_trunk_pair.h "fused_trunk_kernel|fused_trunk_pair_kernel" 10000 > "$OUT/pmc_mnist.json" 2> "$OUT/pmc_mnist.err";;
    traffic) python tools/pmc_traffic.py /tmp/pmc_traffic_FETCH_SIZE /tmp/pmc_traffic_WRITE_SIZE traffic 8 fused_stage.hip,conv_nhwc.hip,conv.hip,trunk.hip,scorer.hip,scan_fast.hip,scan_common.h,logits.hip > "$OUT/pmc_traffic_signs.json" 2> "$OUT/pmc_traffic_signs.err";;
    native50) python tools/pmc_traffic.py /tmp/pmc_native50_FETCH_SIZE /tmp/pmc_native50_WRITE_SIZE native50 8 fused_stage.hip,conv_nhwc.hip,conv.hip,trunk.hip,scorer.hip,scan_fast.hip,scan_common.h,logits.hip > "$OUT/pmc_native50.json" 2> "$OUT/pmc_native50.err";;
    cam) python tools/pmc_traffic.py /tmp/pmc_cam_FETCH_SIZE /tmp/pmc_cam_WRITE_SIZE cam_parts 8 conv_nhwc.hip,aggregate.hip,scorer.hip,scan_cam.hip,scan_common.h,logits.hip > "$OUT/pmc_cam_parts.json" 2> "$OUT/pmc_cam_parts.err";;
  esac
done
cp profiles/pmc_traffic.json "$OUT/pmc_traffic.json"
