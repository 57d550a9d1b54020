#!/bin/bash
# Per-kernel times of the C3 configuration (600 s of 8-channel 24-bit / 96 kHz at -8 -e -p): kernel trace with statistics,
# copied to profiles/<tag>_c3_kernel_stats.csv.
set -u
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
TAG="${1:-r2}"
OUT="$ROOT/gpurun_out/c3trace"
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o t -- python3 "$ROOT/tools/c3_only.py" 600 > "$OUT.log" 2>&1 || { echo "trace failed"; tail -5 "$OUT.log"; exit 1; }
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/**/t_kernel_stats.csv", recursive=True)[0]
rows = list(csv.reader(open(f)))
keep = [rows[0]] + [r for r in rows[1:] if r[0].replace("void ", "").startswith("fa_")]
csv.writer(open("$ROOT/gpurun_out/${TAG}_c3_kernel_stats.csv", "w", newline="")).writerows(keep)
for r in keep[1:]:
    print("%-60s calls %4s avg %10.1f us" % (r[0].split("(")[0][:60], r[1], float(r[3]) / 1e3))
PY
