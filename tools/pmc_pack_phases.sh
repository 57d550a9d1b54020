#!/bin/bash
# Development aid: vector instruction counts of the register-resident pack kernel with parts of it switched off
# (FLACAMD_DEBUG 0x10: no token emission, 0x20: no CRC, 0x40: no subframe heads; 0x8000000, a bit no kernel looks at, is set in
# every pass so that all of them run the same kernel variant; the frames such runs write are wrong, only the counts are of use).
set -u
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/pmc_pack"
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for dbg in 0x8000000 0x8000010 0x8000030 0x8000070; do
	export FLACAMD_DEBUG=$dbg
	timeout -s KILL 240 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d "$OUT/d$dbg" -o p -- python3 "$ROOT/bench.py" --no-cpu-baseline --no-decode --no-configs --no-api --no-sharded --steps 1 --warmup 1 > "$OUT/d$dbg.log" 2>&1 || { echo "pass $dbg failed"; tail -5 "$OUT/d$dbg.log"; }
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
for dbg in ("0x8000000", "0x8000010", "0x8000030", "0x8000070"):
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for fn in glob.glob(sys.argv[1] + "/d" + dbg + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            per[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in sorted(per.items()):
        if "pack_fast" in k:
            w = sum(cs["SQ_WAVES"]) / len(cs["SQ_WAVES"])
            print("debug", dbg, k, "waves %.0f" % w, {c: round(sum(v) / len(v) / w, 1) for c, v in sorted(cs.items()) if c != "SQ_WAVES"})
PY
