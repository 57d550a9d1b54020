#!/bin/bash
# Development aid: per-kernel times of the decode leg at batch sizes up to n hours (kernel trace with statistics).
set -u
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/c4trace"
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o t -- python3 "$ROOT/tools/c4_only.py" "${1:-8}" > "$OUT.log" 2>&1 || { echo "trace failed"; tail -5 "$OUT.log"; exit 1; }
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/**/t_kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if r["Name"].replace("void ", "").startswith("fd_"):
        print("%-44s calls %4s avg %10.1f us  total %10.1f us" % (r["Name"].split("(")[0][:44], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3))
PY
