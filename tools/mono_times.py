#!/usr/bin/env python3
"""Development aid: stage times of one-channel streams (16-bit at -8 and -5, 24-bit at -8) per hour of audio."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flac_amd import engine, synth

minutes = int(sys.argv[1]) if len(sys.argv) > 1 else 20


def run(name, pcm, bps, rate, level, fmt):
    n = len(pcm)
    F = (n + 4095) // 4096
    p = engine.make_params(1, bps, rate, level=level)
    enc = engine.Encoder(p, max_frames=F)
    enc.set_timing(True)
    d = torch.from_numpy(np.ascontiguousarray(pcm)).cuda()
    cap = enc.max_output_bytes(F)
    out = torch.empty(cap, dtype=torch.uint8, device="cuda")
    fb = torch.empty(F, dtype=torch.int32, device="cuda")
    for _ in range(3):
        enc.encode_device(d.data_ptr(), fmt, n, out.data_ptr(), cap, fb.data_ptr())
    t = enc.last_timing()
    print("%-28s %6.2f ms per hour  %s" % (name, sum(t.values()) * 60.0 / minutes, {k: round(v * 60.0 / minutes, 3) for k, v in t.items()}), flush=True)
    enc.close()


m16 = np.tile(synth.music_like(60.0, seed=1)[:, :1].astype(np.int16), (minutes, 1))
run("mono 16-bit, -8", m16, 16, 44100, 8, engine.PCM_S16)
run("mono 16-bit, -5", m16, 16, 44100, 5, engine.PCM_S16)
m24 = np.tile(synth.multichannel_24(60.0, rate=48000, channels=1, seed=5).astype(np.int32), (minutes, 1))
run("mono 24-bit, -8", m24, 24, 48000, 8, engine.PCM_S32)
