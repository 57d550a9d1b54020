#!/usr/bin/env python3
"""Development aid: the C3 configuration alone (flac -8 -e -p, 8 channels, 24-bit / 96 kHz) for profiling runs.
Usage: c3_only.py [seconds]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flac_amd import engine, synth

seconds = int(sys.argv[1]) if len(sys.argv) > 1 else 120
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
print("C3  -8 -e -p 8ch 24/96, %d s: %.1f ms, %.1f M wide samples/s, ratio %.3f" % (n // 96000, dt * 1e3, n / dt / 1e6, total / (n * 24)))
print("    stages ms:", {k: round(v, 2) for k, v in enc.last_timing().items()})
enc.close()
