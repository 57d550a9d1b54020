#!/usr/bin/env python3
"""Development aid: one format group of short clips (16-bit mono by default) through flacamd_encode_streams_device (no MD5) and
flacamd_encode_clips_device (MD5 beside the coding stages), with the stage times of each."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flac_amd import engine, synth

channels = int(sys.argv[1]) if len(sys.argv) > 1 else 1
nclips = int(sys.argv[2]) if len(sys.argv) > 2 else 1250
rng = np.random.default_rng(5)
N = 4096
lens = rng.integers(44100, 30 * 44100, nclips).astype(np.uint64)
frames = (lens + N - 1) // N
starts = np.concatenate([[0], np.cumsum(frames)[:-1]]).astype(np.uint64) * N
F = int(frames.sum())
base = synth.music_like(60.0, seed=1)[:, :channels].astype(np.int16)
pcm = np.tile(base, (F * N // len(base) + 1, 1))[: F * N]
p = engine.make_params(channels, 16, 44100, level=5)
enc = engine.Encoder(p, max_frames=F)
enc.set_timing(True)
d = torch.from_numpy(np.ascontiguousarray(pcm)).cuda()
cap = enc.max_output_bytes(F) + 86 * nclips
out = torch.empty(cap, dtype=torch.uint8, device="cuda")
fb = torch.empty(F, dtype=torch.int32, device="cuda")
print("clips", nclips, "frames", F, "hours %.2f" % (F * N / 44100 / 3600))
for it in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    enc.encode_streams_device(d.data_ptr(), engine.PCM_S16, starts, lens, out.data_ptr(), cap, fb.data_ptr())
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("streams (no MD5): %.2f ms" % (dt * 1e3), {k: round(v, 2) for k, v in enc.last_timing().items()})
for it in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    enc.encode_clips_device(d.data_ptr(), engine.PCM_S16, starts, lens, out.data_ptr(), cap)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("clips (MD5 beside): %.2f ms" % (dt * 1e3), {k: round(v, 2) for k, v in enc.last_timing().items()})
dig = torch.zeros(nclips * 16, dtype=torch.uint8, device="cuda")
for it in range(2):
    enc.use_digests(dig.data_ptr())
    torch.cuda.synchronize(); t0 = time.perf_counter()
    enc.encode_clips_device(d.data_ptr(), engine.PCM_S16, starts, lens, out.data_ptr(), cap)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("clips (digests given, no MD5 kernel): %.2f ms" % (dt * 1e3), {k: round(v, 2) for k, v in enc.last_timing().items()})
