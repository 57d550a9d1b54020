#!/usr/bin/env python3
"""Development aid: per-phase shader-clock deltas of the register-resident search kernel,
from the timestamps it leaves when FLACAMD_DEBUG has bit 0x1000 set."""
import os
import sys

os.environ["FLACAMD_DEBUG"] = hex(0x1000 | int(sys.argv[2], 0) if len(sys.argv) > 2 else 0x1000)
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flac_amd import engine, synth

minutes = int(sys.argv[1]) if len(sys.argv) > 1 else 20
pcm = np.tile(synth.music_like(60, seed=1234).astype(np.int16), (minutes, 1))
n = len(pcm)
F = (n + 4095) // 4096
enc = engine.Encoder(engine.make_params(2, 16, 44100, level=8), max_frames=F)
d = torch.from_numpy(pcm).cuda()
cap = enc.max_output_bytes(F)
out = torch.empty(cap, dtype=torch.uint8, device="cuda")
fb = torch.empty(F, dtype=torch.int32, device="cuda")
for _ in range(2):
    try:
        enc.encode_device(d.data_ptr(), 1, n, out.data_ptr(), cap, fb.data_ptr())
    except engine.EngineError:
        pass
t = enc.debug(5, np.uint64, 131072).reshape(-1, 8).astype(np.int64)
t = t[: min(len(t), F * 4)]
names = ["loads+table -> barrier", "MAC", "MAC barrier wait", "merges", "eval", "wave0 argmin", "wave0 final"]
dt = np.diff(t, axis=1)
ok = (dt >= 0).all(axis=1) & (dt < 10_000_000).all(axis=1)
print("workgroups sampled:", int(ok.sum()), "of", len(t))
for i, nm in enumerate(names):
    print("  %-26s mean %8.0f  median %8.0f  p90 %8.0f ticks" % (nm, dt[ok, i].mean(), np.median(dt[ok, i]), np.percentile(dt[ok, i], 90)))
tot = t[ok, 7] - t[ok, 0]
print("  %-26s mean %8.0f  median %8.0f" % ("whole workgroup", tot.mean(), np.median(tot)))
span = t[ok].max() - t[ok].min()
print("  span of sampled WGs: %d ticks" % span)
sig = np.arange(len(t)) % 4
for k, nm in enumerate(["left", "right", "mid", "side"]):
    m = ok & (sig == k)
    print("  signal %-5s MAC mean %8.0f  whole %8.0f" % (nm, dt[m, 1].mean(), tot[m[ok]].mean() if False else (t[m, 7] - t[m, 0]).mean()))
