#!/bin/bash
# Development aid: the decoder object's rate at several batch sizes (frames decoded per device pass).
set -u
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$ROOT"
python3 - <<PY
import numpy as np, sys
sys.path.insert(0, "$ROOT")
import bench
base, _ = bench.build_workload(60, 1)
base.astype(np.int16).tofile("/dev/shm/api_base.raw")
PY
for b in default 2048 8192 16384; do
  if [ "$b" = default ]; then unset API_RATE_DEC_BATCH; else export API_RATE_DEC_BATCH=$b; fi
  echo "decoder batch $b: $(timeout -k 10 200 tools/bin/api_rate /dev/shm/api_base.raw 60 8 2048 1 | tail -n 1 | python3 -c 'import sys,json; d=json.loads(sys.stdin.readline()); print({k: d[k] for k in d if k.startswith("decoder")})')"
done
rm -f /dev/shm/api_base.raw
