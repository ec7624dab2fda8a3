# Counters of the weight-gradient kernel alone (64 -> 64, 3x3 on 8x8 maps, 1,024 patches: the training step's shape):
#   bash tools/wgrad_pmc.sh      -> per-counter averages of conv_wgrad_taps_kernel
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
cd /tmp && export TMPDIR=/tmp
cat > /tmp/wgrad_probe.py <<PY
import sys, torch
sys.path.insert(0, "$ROOT")
from ips_amd import hip
dev = torch.device("cuda:0")
x = torch.randn((1024, 64, 8, 8), device=dev).contiguous(memory_format=torch.channels_last)
dy = torch.randn((1024, 64, 8, 8), device=dev).contiguous(memory_format=torch.channels_last)
for _ in range(6):
    hip.conv2d_nhwc_wgrad(x, dy, (64, 64, 3, 3), 1, 1)
torch.cuda.synchronize()
PY
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD" "GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  rm -rf /tmp/wg_pmc
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/wg_pmc -o pmc -- python3 /tmp/wgrad_probe.py > /dev/null 2>&1
  python3 - <<'PY'
import csv, glob
from collections import defaultdict
f = glob.glob("/tmp/wg_pmc/**/*counter_collection.csv", recursive=True)
acc = defaultdict(list)
for r in csv.DictReader(open(f[0])) if f else []:
    if "wgrad_taps" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print("%-28s %14.0f  (%d launches)" % (k, sum(v[1:]) / max(len(v) - 1, 1), len(v)))
PY
done
