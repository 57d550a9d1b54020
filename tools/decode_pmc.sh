#!/bin/bash
# Development aid: instruction mix of the decoder kernels (one counter pass over bench.py with the decode leg).
set -u
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/dpmc"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 240 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d "$OUT" -o p -- python3 "$ROOT/bench.py" --no-cpu-baseline --no-configs --no-api --steps 1 --warmup 1 > "$OUT.log" 2>&1 || { echo "pmc pass failed"; tail -5 "$OUT.log"; exit 1; }
python3 - <<EOF
import csv, glob, collections
f = glob.glob("$OUT/**/p_counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if not k.startswith("fd_"): continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVES": n[k] += 1
import json
out = {"command": "bench.py --no-cpu-baseline --no-configs --no-api --steps 1 --warmup 1 (decode leg: one hour of 16/44.1 stereo)", "note": "averages per wave; the decode kernel's two waves per workgroup (parser, restorer) are averaged together; SQ_WAVE_CYCLES and the SQ_WAIT counters are in quad-cycles", "kernels": {}}
for k, v in acc.items():
    launches = n[k]
    w = v["SQ_WAVES"] / launches
    print(k, "launches", launches, "waves/launch %.0f" % w, " per wave:", " ".join("%s=%.0f" % (c.replace("SQ_", ""), v[c] / launches / w) for c in sorted(v) if c != "SQ_WAVES"))
    out["kernels"][k] = {"launches": launches, "waves_per_launch": int(round(w)), "per_wave": {c: round(v[c] / launches / w, 1) for c in sorted(v) if c != "SQ_WAVES"}}
json.dump(out, open("$ROOT/gpurun_out/decode_pmc_summary.json", "w"), indent=1, sort_keys=True)
EOF
