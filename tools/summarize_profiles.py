#!/usr/bin/env python3
"""Turns the rocprofv3 output of tools/profile_round.sh into the committed summaries:
profiles/<tag>_kernel_stats.csv, profiles/<tag>_pmc_per_kernel.csv and profiles/<tag>_pmc_summary.json
(HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE, both in KiB per dispatch as rocprofv3 reports them;
the doubling is the measured gfx950 behaviour for wide coalesced reads)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r1"
minutes = int(sys.argv[2]) if len(sys.argv) > 2 else 60
# optional: the directory under gpurun_out/ to read (tools/c3_profile.sh writes prof_c3) and a name part for the outputs
src = os.path.join(ROOT, "gpurun_out", sys.argv[3] if len(sys.argv) > 3 else "prof_round")
part = sys.argv[4] if len(sys.argv) > 4 else ""
tag = tag + ("_" + part if part else "")


def short(name):
    name = name.replace("void ", "")
    return name.split("(")[0].strip()


VALU_PEAK = 1024 * 2.4e9 / 2.0   # wave-instructions per second: 1024 SIMDs, one full-rate wave64 VALU op per 2 cycles at 2.4 GHz
avg_ns = {}
stats = glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    rows = list(csv.reader(open(stats[0])))
    for r in csv.DictReader(open(stats[0])):
        avg_ns[short(r["Name"])] = float(r["AverageNs"])
    with open(os.path.join(ROOT, "profiles", tag + "_kernel_stats.csv"), "w", newline="") as f:
        csv.writer(f).writerows(rows)
    print("kernel stats:", stats[0], len(rows) - 1, "kernels")

per = defaultdict(lambda: defaultdict(list))
for d in glob.glob(os.path.join(src, "pmc_*")):
    if not os.path.isdir(d):
        continue
    for fn in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(fn)):
            per[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
counters = sorted({c for k in per.values() for c in k})
with open(os.path.join(ROOT, "profiles", tag + "_pmc_per_kernel.csv"), "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "dispatches"] + ["mean_" + c for c in counters])
    for k, cs in sorted(per.items()):
        n = max(len(v) for v in cs.values())
        w.writerow([k, n] + [("%.1f" % (sum(cs[c]) / len(cs[c])) if c in cs else "") for c in counters])
summary = {"minutes": minutes, "workload": open(os.path.join(src, "command.txt")).read().strip() if os.path.exists(os.path.join(src, "command.txt")) else "bench.py, one hour of 16-bit / 44.1 kHz stereo per step", "note": "per launch; FETCH_SIZE/WRITE_SIZE in KiB as reported, hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024; valu_issue_frac = SQ_INSTS_VALU / time / (1024 SIMDs * 2.4 GHz / 2)", "kernels": {}}
for k, cs in per.items():
    if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
        fe, wr = sum(cs["FETCH_SIZE"]) / len(cs["FETCH_SIZE"]), sum(cs["WRITE_SIZE"]) / len(cs["WRITE_SIZE"])
        e = {"fetch_size_kib": round(fe, 1), "write_size_kib": round(wr, 1), "hbm_bytes_per_launch": int((2 * fe + wr) * 1024)}
        if "SQ_INSTS_VALU" in cs and k in avg_ns:
            insts = sum(cs["SQ_INSTS_VALU"]) / len(cs["SQ_INSTS_VALU"])
            e["avg_ms"] = round(avg_ns[k] / 1e6, 4)
            e["valu_wave_insts_per_launch"] = int(insts)
            # share of the chip's full-rate vector issue slots the kernel fills (half-rate integer ops
            # such as v_mad_i32_i24 / v_dot2 / v_sad occupy two slots but count once: a lower bound on busy)
            e["valu_issue_frac"] = round(insts / (avg_ns[k] * 1e-9) / VALU_PEAK, 4)
        summary["kernels"][k] = e
json.dump(summary, open(os.path.join(ROOT, "profiles", tag + "_pmc_summary.json"), "w"), indent=1, sort_keys=True)
print(json.dumps(summary, indent=1, sort_keys=True))
