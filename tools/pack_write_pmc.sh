#!/bin/bash
# Development aid: bytes written per kernel of the encode step (one WRITE_SIZE pass over bench.py).
set -u
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/wpmc"
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for pass in WRITE_SIZE FETCH_SIZE; do
timeout -s KILL 240 rocprofv3 --pmc $pass --output-format csv -d "$OUT/$pass" -o p -- python3 "$ROOT/bench.py" --no-cpu-baseline --no-decode --no-configs --no-api --steps 2 --warmup 1 > "$OUT/$pass.log" 2>&1 || { echo "pass failed"; tail -5 "$OUT/$pass.log"; exit 1; }
done
python3 - <<PY
import csv, glob, collections
for name in ("WRITE_SIZE", "FETCH_SIZE"):
    f = glob.glob("$OUT/%s/**/p_counter_collection.csv" % name, recursive=True)[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()):
        if k.startswith("fa_"): print(name, k, "MiB per launch %.1f" % (sum(v) / len(v) / 1024))
PY
