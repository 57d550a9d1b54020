#!/bin/bash
# Vector instruction counts of the search kernel with its phases switched off one at a time (FLACAMD_DEBUG bit 0: no
# multiply-add phase, bit 1: no evaluation phase; the frames such a run writes are wrong, only the counts are of use).
set -u
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/pmc_phases"
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for dbg in 0 1 2 3; do
	export FLACAMD_DEBUG=$dbg
	timeout -s KILL 240 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d "$OUT/d$dbg" -o p -- python3 "$ROOT/bench.py" --no-cpu-baseline --no-decode --no-configs --no-api --steps 1 --warmup 1 > "$OUT/d$dbg.log" 2>&1 || { echo "pass $dbg failed"; tail -5 "$OUT/d$dbg.log"; }
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
for dbg in "0123":
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for fn in glob.glob(sys.argv[1] + "/d" + dbg + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            per[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in sorted(per.items()):
        if "search_fast" in k or "pack_fast" in k:
            print("debug", dbg, k, {c: round(sum(v) / len(v) / 1e6, 2) for c, v in sorted(cs.items())})
PY
