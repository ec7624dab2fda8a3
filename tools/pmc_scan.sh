# rocprofv3 counters of the selection loop alone (tools/scan_stamps.py cam): dynamic instruction counts and busy cycles per launch
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
cd /tmp && export TMPDIR=/tmp
cd "$ROOT"
out=${1:-gpurun_out/pmc_scan}
mkdir -p $out
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES -d $out/pmc1 -o pmc1 --output-format csv -- python3 tools/scan_stamps.py cam > $out/pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INST_CYCLES_VALU SQ_ACTIVE_INST_VALU -d $out/pmc2 -o pmc2 --output-format csv -- python3 tools/scan_stamps.py cam > $out/pmc2.log 2>&1
python3 - $out <<'PY'
import csv, collections, sys
out = sys.argv[1]
for f in (out + "/pmc1/pmc1_counter_collection.csv", out + "/pmc2/pmc2_counter_collection.csv"):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "scan_" in r["Kernel_Name"] and "<true" not in r["Kernel_Name"]:
            agg[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k, v in sorted(agg.items()):
        print("%-60s %-22s launches %2d  per launch %12.0f  per iteration (255) %9.1f" % (k[0][-60:], k[1], len(v), sum(v) / len(v), sum(v) / len(v) / 255))
PY
