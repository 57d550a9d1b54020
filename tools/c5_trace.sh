#!/bin/bash
# Development aid: per-kernel times of the clip-batch configuration of bench.py.
set -u
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/c5trace"
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o t -- python3 "$ROOT/tools/c5_only.py" > "$OUT.log" 2>&1 || { echo "trace failed"; tail -5 "$OUT.log"; exit 1; }
tail -2 "$OUT.log"
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/**/t_kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print("%-60s calls %5s total %9.2f ms avg %9.1f us" % (r["Name"].split("(")[0][:60], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
PY
