#!/bin/bash
# Development aid: vector instruction counts of the general search kernel on the 8-channel 24-bit -8 -e -p configuration
# with its phases switched off one at a time (FLACAMD_DEBUG bit 0: no multiply-add phase, bit 1: no evaluation phase;
# the frames such a run writes are wrong, only the counts are of use).
set -u
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/c3_phases"
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for dbg in 0 1 2 3; do
	export FLACAMD_DEBUG=$dbg
	timeout -s KILL 240 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM --output-format csv -d "$OUT/d$dbg" -o p -- python3 "$ROOT/tools/c3_only.py" 120 > "$OUT/d$dbg.log" 2>&1 || { echo "pass $dbg failed"; tail -5 "$OUT/d$dbg.log"; }
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
for dbg in "0123":
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for fn in glob.glob(sys.argv[1] + "/d" + dbg + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            per[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in sorted(per.items()):
        if "fa_search" in k:
            w = sum(cs["SQ_WAVES"]) / len(cs["SQ_WAVES"])
            print("debug", dbg, k, "waves %.0f" % w, {c: round(sum(v) / len(v) / w, 1) for c, v in sorted(cs.items()) if c != "SQ_WAVES"})
PY
