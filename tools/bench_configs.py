#!/usr/bin/env python3
"""Stage times of the secondary BASELINE configurations (not the headline bench line):
C3 = flac -8 -e -p on 24-bit / 96 kHz / 8 channels, C5 = batch of short clips at -5."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flac_amd import clips, engine, synth

seconds = int(sys.argv[1]) if len(sys.argv) > 1 else 60
base = synth.multichannel_24(10.0, rate=96000, channels=8, seed=4321)
pcm = np.tile(base, (max(1, seconds // 10), 1))
n = len(pcm)
F = (n + 4095) // 4096
p = engine.make_params(8, 24, 96000, level=8, do_exhaustive_model_search=1, do_qlp_coeff_prec_search=1)
enc = engine.Encoder(p, max_frames=F)
enc.set_timing(True)
d = torch.from_numpy(pcm).cuda()
cap = enc.max_output_bytes(F)
out = torch.empty(cap, dtype=torch.uint8, device="cuda")
fb = torch.empty(F, dtype=torch.int32, device="cuda")
for it in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    total = enc.encode_device(d.data_ptr(), engine.PCM_S32, n, out.data_ptr(), cap, fb.data_ptr())
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
print("C3  -8 -e -p 8ch 24/96, %d s: %.1f ms, %.1f M wide samples/s (%.1f M channel samples/s), ratio %.3f" % (n // 96000, dt * 1e3, n / dt / 1e6, 8 * n / dt / 1e6, total / (n * 24)))
print("    stages ms:", {k: round(v, 2) for k, v in enc.last_timing().items()})
enc.close()
del d, out

rng = np.random.default_rng(777)
batch = []
nclips = int(sys.argv[2]) if len(sys.argv) > 2 else 200
for i in range(nclips):
    dur = float(rng.uniform(1.0, 30.0))
    bps = 16 if rng.random() < 0.5 else 24
    rate = 44100 if rng.random() < 0.5 else 48000
    ch = 1 if rng.random() < 0.5 else 2
    src = synth.music_like(30.0, rate=rate, seed=5 + (i % 4))[:, :ch] if bps == 16 else synth.multichannel_24(30.0, rate=rate, channels=ch, seed=9 + (i % 4))
    batch.append((np.ascontiguousarray(src[: int(dur * rate)]), bps, rate))
tot = sum(len(c[0]) for c in batch)
clips.encode_clips(batch[:8], level=5)
t0 = time.perf_counter()
res = clips.encode_clips(batch, level=5)
dt = time.perf_counter() - t0
print("C5  %d clips at -5 (%.0f s of audio): %.1f ms wall incl. upload and download, %.1f M wide samples/s, %.0f clips/s" % (nclips, sum(len(c[0]) / c[2] for c in batch), dt * 1e3, tot / dt / 1e6, nclips / dt))
