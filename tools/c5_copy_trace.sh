#!/bin/bash
# Development aid: kernels and memory copies of the clip-batch configuration on one time line.
set -u
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/c5copy"
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 400 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d "$OUT" -o t -- python3 "$ROOT/tools/c5_only.py" > "$OUT.log" 2>&1 || { echo "trace failed"; tail -5 "$OUT.log"; exit 1; }
ls "$OUT"
