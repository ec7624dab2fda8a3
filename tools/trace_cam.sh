cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04k
export IPSX_CAM_SHORT=-20 IPSX_CAM_WGS=255
rocprofv3 --kernel-trace -d gpurun_out/r04k/trace -o t --output-format csv -- python3 bench.py --config cam --cpu-seconds 0 --steps 5 --warmup 3 --no-kernel-events > gpurun_out/r04k/bench.json 2> gpurun_out/r04k/bench.err
python3 tools/timeline.py gpurun_out/r04k/trace 40 | tee gpurun_out/r04k/timeline.txt
rm -rf gpurun_out/r04k/trace
