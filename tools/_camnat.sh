for w in -11 -19 -27 -35 -2 -12 -20; do
echo "== IPSX_CAM_SHORT=$w"
IPSX_CAM_SHORT=$w python bench.py --config cam_native --cpu-seconds 0 --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('value %.2f M  ms/step %.4f slots %s  stream %.4f' % (d['value']/1e6, d['ms_per_step'], d['parity'].get('slots_equal'), d['roofline']['launch_ms']))
"
done
