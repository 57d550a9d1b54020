#!/bin/bash
# Development aid: per-kernel times of the one-channel configurations (kernel trace with statistics over tools/mono_times.py).
set -u
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/monotrace"
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o t -- python3 "$ROOT/tools/mono_times.py" "${1:-60}" > "$OUT.log" 2>&1 || { echo "trace failed"; tail -5 "$OUT.log"; exit 1; }
tail -n 3 "$OUT.log"
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/**/t_kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if r["Name"].replace("void ", "").startswith("fa_"):
        print("%-60s calls %4s avg %10.1f us" % (r["Name"].split("(")[0][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
