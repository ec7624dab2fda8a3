#!/bin/bash
# The evidence for the loop beyond the LDS as a team of workgroups (csrc/scan_large_team.h) in one GPU call:
#   bash tools/collect_team_profiles.sh <out dir under gpurun_out>
set -u
cd "$(dirname "$0")/.."
O=${1:-gpurun_out/team}
mkdir -p "$O"
export TMPDIR=/tmp
{ echo "# tools/scan_stamps.py large: the main workgroup of a team of 8 (scan_large_team_kernel), then one workgroup (scan_large_kernel)"; TEAM=8 python tools/scan_stamps.py large 2>&1 | grep -v amdgpu.ids; echo; TEAM=0 python tools/scan_stamps.py large 2>&1 | grep -v amdgpu.ids; } > "$O/scan_stamps_large.txt"
{ echo "# tools/scan_stamps.py largepipe: the team loop inside ips() at the shipped CAMELYON sizes"; python tools/scan_stamps.py largepipe 2>&1 | grep -v amdgpu.ids; } > "$O/scan_stamps_largepipe.txt"
python tools/scan_team_check.py 2>&1 | grep -v amdgpu.ids > "$O/scan_team_check.txt"
python tools/scan_team_check.py fuzz 300 20000 2>&1 | grep -v amdgpu.ids > "$O/scan_team_fuzz.txt"
{ LARGE=1 python tools/fuzz_pipelines.py 0 24 2>&1 | tail -1; } > "$O/fuzz_pipelines_large.txt"
python bench.py --config cam_native --cpu-seconds 8 2>/dev/null > "$O/bench_cam_native.json"
for t in 0 2 4 8; do echo "IPSX_LARGE_TEAM=$t: $(IPSX_LARGE_TEAM=$t python bench.py --config cam_native --cpu-seconds 0 --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.2f M patches/s, %.4f ms per call, stream kernel %.4f ms, slots equal %s' % (d['value']/1e6, d['ms_per_step'], d['roofline']['launch_ms'], d['parity'].get('slots_equal')))")"; done > "$O/cam_native_team_widths.txt"
python tools/soak.py 120 40 cam_native,cam 2>&1 | tail -12 > "$O/soak_team.txt"
