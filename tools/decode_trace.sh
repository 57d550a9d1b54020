#!/bin/bash
# Development aid: per-kernel times of the decode leg of bench.py (kernel trace with statistics).
set -u
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/dtrace"
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o t -- python3 "$ROOT/bench.py" --no-cpu-baseline --no-configs --no-api --steps 3 --warmup 1 > "$OUT.log" 2>&1 || { echo "trace failed"; tail -5 "$OUT.log"; exit 1; }
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/**/t_kernel_stats.csv", recursive=True)[0]
rows = list(csv.reader(open(f)))
keep = [rows[0]] + [r for r in rows[1:] if r[0].replace("void ", "").startswith("fd_")]
csv.writer(open("$ROOT/gpurun_out/decode_kernel_stats.csv", "w", newline="")).writerows(keep)
for r in csv.DictReader(open(f)):
    if r["Name"].replace("void ", "").startswith("fd_"):
        print("%-40s calls %4s avg %10.1f us" % (r["Name"].split("(")[0][:40], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
