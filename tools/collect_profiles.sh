#!/bin/bash
# Round evidence in one GPU call: every bench line, rocprofv3 kernel statistics of the headline / cam / native50 commands,
# HBM traffic from separate --pmc passes (FETCH_SIZE, WRITE_SIZE; tools/pmc_traffic.py applies the gfx950 correction and
# updates profiles/pmc_traffic.json), the loop stamps, and the parity rates the tests print.
#   bash tools/collect_profiles.sh <out dir under gpurun_out>
set -u
cd "$(dirname "$0")/.."
OUT=${1:-gpurun_out/profiles}
mkdir -p "$OUT"
export TMPDIR=/tmp
bash tools/bench_all.sh > "$OUT/bench_all.jsonl" 2> "$OUT/bench_all.err"
for c in mnist cam native50 cam_native; do
  rm -rf /tmp/ks_$c
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$c -o ks -- python3 "$OLDPWD/bench.py" --config $c --steps 10 --warmup 3 --cpu-seconds 0 > "$OLDPWD/$OUT/bench_${c}_profiled.json" 2>/dev/null)
  cp $(find /tmp/ks_$c -name "*kernel_stats.csv" | head -1) "$OUT/${c}_kernel_stats.csv" 2>/dev/null
done
for ctr in FETCH_SIZE WRITE_SIZE; do            # the projector stream alone (inside ips() counter collection would serialise it away)
  rm -rf /tmp/pmc_stream_$ctr
  (cd /tmp && rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/pmc_stream_$ctr -o pmc -- python3 "$OLDPWD/tools/projector_stream_bench.py" pmc > /dev/null 2>&1)
done
for c in mnist cam native50 traffic; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_${c}_$ctr
    (cd /tmp && rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/pmc_${c}_$ctr -o pmc -- python3 "$OLDPWD/bench.py" --config $c --steps 3 --warmup 1 --cpu-seconds 0 > /dev/null 2>&1)
  done
done
cp profiles/pmc_traffic.json "$OUT/pmc_traffic_before.json"
python tools/pmc_traffic.py /tmp/pmc_mnist_FETCH_SIZE /tmp/pmc_mnist_WRITE_SIZE mnist 8 fused_trunk.hip,fused_trunk_pair.h "fused_trunk_kernel|fused_trunk_pair_kernel" 10000 > "$OUT/pmc_mnist.json" 2> "$OUT/pmc_mnist.err"
python tools/pmc_traffic.py /tmp/pmc_cam_FETCH_SIZE /tmp/pmc_cam_WRITE_SIZE cam_parts 8 conv_nhwc.hip,aggregate.hip,scorer.hip,scan_cam.hip,scan_common.h,logits.hip > "$OUT/pmc_cam_parts.json" 2> "$OUT/pmc_cam_parts.err"
python tools/pmc_traffic.py /tmp/pmc_stream_FETCH_SIZE /tmp/pmc_stream_WRITE_SIZE cam 5 conv_nhwc.hip,ipsx_rowstats.h "projector_stream_kernel" 65536 > "$OUT/pmc_cam.json" 2> "$OUT/pmc_cam.err"
python tools/pmc_traffic.py /tmp/pmc_traffic_FETCH_SIZE /tmp/pmc_traffic_WRITE_SIZE traffic 8 fused_stage.hip,conv_nhwc.hip,conv.hip,trunk.hip,scorer.hip,scan_fast.hip,scan_common.h,logits.hip > "$OUT/pmc_traffic_signs.json" 2> "$OUT/pmc_traffic_signs.err"
python tools/pmc_traffic.py /tmp/pmc_native50_FETCH_SIZE /tmp/pmc_native50_WRITE_SIZE native50 8 fused_stage.hip,conv_nhwc.hip,conv.hip,trunk.hip,scorer.hip,scan_fast.hip,scan_common.h,logits.hip > "$OUT/pmc_native50.json" 2> "$OUT/pmc_native50.err"
cp profiles/pmc_traffic.json "$OUT/pmc_traffic.json"
python tools/scan_compare.py > "$OUT/scan_compare.txt" 2>&1
python tools/scan_stamps.py camwaves > "$OUT/scan_camwaves.txt" 2>&1
bash tools/pmc_scan.sh "$OUT/pmc_scan" > "$OUT/scan_pmc.txt" 2>&1
IPSX_SCAN_R8=0 bash tools/pmc_scan.sh "$OUT/pmc_scan_fast" >> "$OUT/scan_pmc.txt" 2>&1
rm -rf "$OUT/pmc_scan" "$OUT/pmc_scan_fast"
python tools/train_step_breakdown.py --fused --profile > "$OUT/train_step_hipconv.txt" 2>&1
IPSX_TRAIN_CONV=0 python tools/train_step_breakdown.py --fused --profile > "$OUT/train_step_miopen.txt" 2>&1
python tools/scan_stamps.py cam > "$OUT/scan_stamps_cam.txt" 2>&1
IPSX_SCAN_R8=0 python tools/scan_stamps.py cam >> "$OUT/scan_stamps_cam.txt" 2>&1
python tools/scan_stamps.py mnist >> "$OUT/scan_stamps_cam.txt" 2>&1
bash tools/collect_team_profiles.sh "$OUT"          # scan_stamps_large(pipe), scan_team_check, cam_native widths, soak_team
{ echo; echo "# the same on the logits of the bench workload itself (tools/scan_stamps.py large bench)"; python tools/scan_stamps.py large bench; } >> "$OUT/scan_stamps_large.txt" 2>&1
python tools/scan_stamps.py campipe > "$OUT/scan_stamps_campipe.txt" 2>&1
python tools/trunk_pair_bench.py > "$OUT/trunk_pair_bench.txt" 2>&1
python tools/projector_stream_bench.py > "$OUT/projector_stream_bench.txt" 2>&1
python tools/trunk_stream_bench.py > "$OUT/trunk_stream_bench.txt" 2>&1
python tools/loop_beside.py > "$OUT/loop_beside.txt" 2>&1
python tools/scan_beside.py > "$OUT/scan_beside.txt" 2>&1
python -m pytest tests/test_bench_parity.py tests/test_seed_sweep.py -q -m gpu -s 2>&1 | grep -v "amdgpu.ids" > "$OUT/parity_rates.txt"
# round 5: the default bench line as the driver runs it, kernel timelines of one-image / one-slide calls, the tie orders on the
# shipped CAMELYON sizes, per-layer rates of the 50-px trunk, the compiler's resource table, the stream's launch-time spread, soak
python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
for c in b1 cam cam_native; do bash tools/trace_step.sh $c 12 2>&1 | grep -v "amdgpu.ids" > "$OUT/${c}_timeline.txt"; done
{ for m in torch torch_all canonical; do
    echo "IPSX_TIE_ORDER=$m python bench.py --config cam_native:"
    IPSX_TIE_ORDER=$m python bench.py --config cam_native --cpu-seconds 0 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); p = d['parity']
print('   %.2f M patches/s, %.3f ms per call; same SET as the reference every image: %s, slots_equal %.4f, selected_in_common %.4f' % (d['value'] / 1e6, d['ms_per_step'], p['selected_in_common'] == 1.0, p['slots_equal'], p['selected_in_common']))"
  done; } > "$OUT/cam_native_tie_orders.txt" 2>&1
python tools/conv_layers.py 14400 2>&1 | grep -v "amdgpu.ids" > "$OUT/conv_layers_native50.txt"
python tools/kernel_resources.py > "$OUT/kernel_resources.txt" 2>/dev/null
python tools/stream_outliers.py 40 2>&1 | grep -v "amdgpu.ids" > "$OUT/stream_outliers.txt"
python tools/soak.py 150 20 2>&1 | grep -v "amdgpu.ids" > "$OUT/soak.txt"
# the soak with torch's default CPU pools (the container's CPU quota then freezes the process: DESIGN 6 "Soak"), and fuzzing
# against the oracle / of the schedules against each other
IPSX_SOAK_THREADS=0 python tools/soak.py 240 20 2>&1 | grep -v "amdgpu.ids" > "$OUT/soak_default_threads.txt"
python tools/fuzz_scan.py 1000 30000 4000 2>&1 | grep -v "amdgpu.ids" > "$OUT/fuzz_scan.txt"
python tools/fuzz_e2e.py 100 400 2>&1 | grep -v "amdgpu.ids" > "$OUT/fuzz_e2e.txt"
python tools/fuzz_pipelines.py 0 750 2>&1 | grep -v "amdgpu.ids" > "$OUT/fuzz_pipelines.txt"
# round 6: the bf16 trunk's phases (second build; alone on its unit; the first build), the parity margins, the sharded path
# with more than one rank on this GPU (fp32 and configs[4]: bf16 pipe + fp16-stored patches)
python tools/fused_stamps.py 40000 bf16 2>&1 | grep -v "amdgpu.ids" > "$OUT/fused_stamps_bf16v3.txt"
{ echo "# IPSX_BF16_BUILD=2: the second build (fused_trunk_bf16v2.h)"; IPSX_BF16_BUILD=2 python tools/fused_stamps.py 40000 bf16; } 2>&1 | grep -v "amdgpu.ids" > "$OUT/fused_stamps_bf16v2.txt"
{ echo "# IPSX_BF16_BUILD=2 IPSX_BF16_ONE_WG=1: one workgroup per unit (one wave per SIMD)"; IPSX_BF16_BUILD=2 IPSX_BF16_ONE_WG=1 python tools/fused_stamps.py 40000 bf16; } 2>&1 | grep -v "amdgpu.ids" >> "$OUT/fused_stamps_bf16v2.txt"
{ echo "# IPSX_BF16_BUILD=1: the first build (fused_trunk_split.h)"; IPSX_BF16_BUILD=1 python tools/fused_stamps.py 40000 bf16; } 2>&1 | grep -v "amdgpu.ids" >> "$OUT/fused_stamps_bf16v2.txt"
python tools/trunk_bf16_bench.py 2>&1 | grep -v "amdgpu.ids" > "$OUT/trunk_bf16_bench.txt"
python tools/fused_stamps.py 40000 fp32x3 2>&1 | grep -v "amdgpu.ids" > "$OUT/fused_stamps_x3.txt"
python -m pytest tests/test_parity_margin.py -q -m gpu -s 2>&1 | grep -v "amdgpu.ids" > "$OUT/parity_margin.txt"
for w in 2 4; do for ps in "fp32 f32" "bf16 f16"; do set -- $ps
  python -m torch.distributed.run --nnodes=1 --nproc-per-node $w --master-addr 127.0.0.1 --master-port 2957$w tools/dist_check.py --backend gloo --share-gpu --bench-shape --precision $1 --storage $2 --cases mnist_ragged,mnist_full 2>&1 | grep "rank "
done; done > "$OUT/dist_check.txt"
echo done
