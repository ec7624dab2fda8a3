mkdir -p gpurun_out/r04j
for cfg in "0 0" "-2 248" "-2 252" "-2 255" "-3 248" "-4 248" "-5 248" "-12 248" "-13 248" "-20 248" "-21 248" "-28 248" "-20 252" "-20 255" "-21 255" "-36 255" "-37 255" "-29 255"; do
  set -- $cfg
  r=$(IPSX_CAM_SHORT=$1 IPSX_CAM_WGS=$2 python bench.py --config cam --cpu-seconds 0 --steps 30 --warmup 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,2), round(d['ms_per_step'],3), round(d['value_pipelined']/1e6,2), d['parity']['indices_equal'])")
  echo "short=$1 wgs=$2 -> $r" | tee -a gpurun_out/r04j/sweep.txt
done
