cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06t; mkdir -p $O
python -m pytest tests -q -m gpu -x 2>&1 | tail -4 > $O/pytest_gpu.txt
{ echo "# tools/scan_stamps.py large: the main workgroup of a team of 8 (scan_large_team_kernel), then one workgroup (scan_large_kernel)"; TEAM=8 python tools/scan_stamps.py large 2>&1 | grep -v amdgpu.ids; echo; TEAM=0 python tools/scan_stamps.py large 2>&1 | grep -v amdgpu.ids; } > $O/scan_stamps_large.txt
{ echo "# tools/scan_stamps.py largepipe: the team loop inside ips() at the shipped CAMELYON sizes"; python tools/scan_stamps.py largepipe 2>&1 | grep -v amdgpu.ids; } > $O/scan_stamps_largepipe.txt
python tools/scan_team_check.py 2>&1 | grep -v amdgpu.ids > $O/scan_team_check.txt
python tools/fuzz_scan.py 0 40 2>&1 | tail -3 > $O/fuzz_scan.txt
{ LARGE=1 python tools/fuzz_pipelines.py 0 24 2>&1 | tail -2; python tools/fuzz_pipelines.py 200 42 2>&1 | tail -2; } > $O/fuzz_pipelines.txt
python bench.py --config cam_native --cpu-seconds 8 2>/dev/null > $O/bench_cam_native.json
bash tools/trace_step.sh cam_native 14 2>&1 | grep -v amdgpu.ids > $O/cam_native_timeline.txt
for t in 0 2 4 8; do echo "IPSX_LARGE_TEAM=$t: $(IPSX_LARGE_TEAM=$t python bench.py --config cam_native --cpu-seconds 0 --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.2f M patches/s, %.4f ms per call, stream kernel %.4f ms, slots equal %s' % (d['value']/1e6, d['ms_per_step'], d['roofline']['launch_ms'], d['parity'].get('slots_equal')))")"; done > $O/cam_native_team_widths.txt
tail -3 $O/pytest_gpu.txt; cat $O/cam_native_team_widths.txt; tail -2 $O/fuzz_scan.txt $O/fuzz_pipelines.txt
