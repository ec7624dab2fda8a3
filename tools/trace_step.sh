# kernel timeline of the last calls of a bench.py run:  bash tools/trace_step.sh <config> [rows of timeline] [extra bench args]
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
cd /tmp && export TMPDIR=/tmp
cd "$ROOT"
cfg=${1:-cam}; n=${2:-40}; shift; shift
out=gpurun_out/trace_$cfg
mkdir -p $out
rocprofv3 --kernel-trace -d $out/trace -o t --output-format csv -- python3 bench.py --config $cfg --cpu-seconds 0 --steps 5 --warmup 3 --no-kernel-events "$@" > $out/bench.json 2> $out/bench.err
python3 tools/timeline.py $out/trace $n | tee $out/timeline.txt
rm -rf $out/trace
