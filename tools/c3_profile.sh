#!/bin/bash
# Profiles of the 8-channel 24-bit / 96 kHz -8 -e -p configuration (600 s of audio per launch): kernel-trace
# statistics, then hardware counters in separate passes.  Output under gpurun_out/prof_c3, summarised by
# tools/summarize_profiles.py <tag> 10 prof_c3 c3.
set -u
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/prof_c3"
rm -rf "$OUT"; mkdir -p "$OUT"
echo "tools/c3_only.py 600: 600 s of 8-channel 24-bit / 96 kHz audio at -8 -e -p, one launch of each kernel per encode call" > "$OUT/command.txt"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o t -- python3 "$ROOT/tools/c3_only.py" 600 > "$OUT/trace.log" 2>&1 || { echo "trace pass failed"; tail -5 "$OUT/trace.log"; exit 1; }
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY"; do
	name=$(echo "$pass" | cut -d' ' -f1)
	timeout -s KILL 240 rocprofv3 --pmc $pass --output-format csv -d "$OUT/pmc_$name" -o p -- python3 "$ROOT/tools/c3_only.py" 600 > "$OUT/pmc_$name.log" 2>&1 || { echo "pmc pass $name failed"; tail -5 "$OUT/pmc_$name.log"; exit 1; }
done
tail -2 "$OUT/trace.log"
