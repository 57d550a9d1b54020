#!/bin/bash
# Development aid: what the device does while the decoder object works through an hour (kernel and copy statistics of tools/bin/api_rate).
set -u
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/apidec"
rm -rf "$OUT"; mkdir -p "$OUT"
cd "$ROOT"
python3 - <<PY
import numpy as np, sys
sys.path.insert(0, "$ROOT")
import bench
base, _ = bench.build_workload(60, 1)
base.astype(np.int16).tofile("/dev/shm/api_base.raw")
PY
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d "$OUT" -o t -- "$ROOT/tools/bin/api_rate" /dev/shm/api_base.raw 60 8 2048 1 > "$OUT.log" 2>&1 || { echo "trace failed"; tail -5 "$OUT.log"; exit 1; }
tail -n 1 "$OUT.log" | cut -c1-300
python3 - <<PY
import csv, glob
for name in ("t_kernel_stats.csv", "t_memory_copy_stats.csv"):
    for f in glob.glob("$OUT/**/" + name, recursive=True):
        for r in csv.DictReader(open(f)):
            print("%-50s calls %6s total %10.1f ms avg %9.1f us" % (r["Name"].split("(")[0][:50], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
PY
rm -f /dev/shm/api_base.raw
