# HBM bytes of the projector stream alone (two counter passes), printed - profiles/pmc_traffic.json is NOT touched:
#   bash tools/pmc_stream.sh
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
cd /tmp && export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmcs_$ctr
  rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/pmcs_$ctr -o pmc -- python3 "$ROOT/tools/projector_stream_bench.py" pmc > /dev/null 2>&1
done
python3 - <<'PY'
import csv, glob
out = {}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("/tmp/pmcs_%s/**/*counter_collection.csv" % ctr, recursive=True)[0]
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if r["Counter_Name"] == ctr and "projector_stream" in r["Kernel_Name"]]
    out[ctr] = sum(v) / len(v) * 1024
rd, wr = 2 * out["FETCH_SIZE"], out["WRITE_SIZE"]   # gfx950: FETCH_SIZE reports half of a wide read stream
print("projector stream, per launch: read %.3f GB  written %.3f GB  total %.3f GB" % (rd / 1e9, wr / 1e9, (rd + wr) / 1e9))
PY
