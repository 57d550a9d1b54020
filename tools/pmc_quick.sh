#!/bin/bash
# One SQ counter pass over a short bench.py run; per-kernel means printed.  Output under gpurun_out/pmc_quick.
set -u
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/pmc_quick"
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 240 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d "$OUT/sq" -o p -- python3 "$ROOT/bench.py" --no-cpu-baseline --no-decode --no-configs --no-api --no-sharded --steps 2 --warmup 1 > "$OUT/sq.log" 2>&1 || { echo "pmc pass failed"; tail -5 "$OUT/sq.log"; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
per = collections.defaultdict(lambda: collections.defaultdict(list))
for fn in glob.glob(sys.argv[1] + "/sq/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        per[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in sorted(per.items()):
    if not k.startswith("void fa_") and not k.startswith("fa_"): continue
    print(k, {c: round(sum(v) / len(v) / 1e6, 2) for c, v in sorted(cs.items())})
PY
