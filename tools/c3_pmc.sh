#!/bin/bash
# Development aid: instruction counts of the kernels of the 8-channel 24-bit -e -p configuration.
set -u
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/c3pmc"
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d "$OUT" -o p -- python3 "$ROOT/tools/c3_only.py" 120 > "$OUT.log" 2>&1 || { echo "pmc failed"; tail -5 "$OUT.log"; exit 1; }
python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/**/p_counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if not k.startswith("fa_"): continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVES": n[k] += 1
for k, v in acc.items():
    w = v["SQ_WAVES"] / n[k]
    print(k[:50], "launches", n[k], "waves %.0f" % w, " per wave:", " ".join("%s=%.0f" % (c.replace("SQ_", ""), v[c] / n[k] / w) for c in sorted(v) if c not in ("SQ_WAVES", "SQ_BUSY_CYCLES")))
PY
