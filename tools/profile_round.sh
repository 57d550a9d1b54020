#!/bin/bash
# Profiles of the bench.py command on the GPU box: kernel-trace statistics, then hardware
# counters in separate passes (one block's counters per run).  Output under gpurun_out/.
set -u
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/prof_round"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o t -- python3 "$ROOT/bench.py" --no-cpu-baseline --no-decode --no-configs --no-api --no-sharded --steps 5 --warmup 2 > "$OUT/trace.log" 2>&1 || { echo "trace pass failed"; tail -5 "$OUT/trace.log"; exit 1; }
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY"; do
	name=$(echo "$pass" | cut -d' ' -f1)
	timeout -s KILL 240 rocprofv3 --pmc $pass --output-format csv -d "$OUT/pmc_$name" -o p -- python3 "$ROOT/bench.py" --no-cpu-baseline --no-decode --no-configs --no-api --no-sharded --steps 2 --warmup 1 > "$OUT/pmc_$name.log" 2>&1 || { echo "pmc pass $name failed"; tail -5 "$OUT/pmc_$name.log"; exit 1; }
done
find "$OUT" -name "*.csv" | head -20
