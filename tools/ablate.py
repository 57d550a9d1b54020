#!/usr/bin/env python3
"""Development aid: stage times of the -8 workload under FLACAMD_DEBUG ablation masks
(each mask in its own process: the mask is read when the encoder is created)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, numpy as np, torch
sys.path.insert(0, %r)
from flac_amd import engine, synth
minutes = int(sys.argv[1])
pcm = np.tile(synth.music_like(60, seed=1234).astype(np.int16), (minutes, 1))
n = len(pcm); F = (n + 4095) // 4096
enc = engine.Encoder(engine.make_params(2, 16, 44100, level=int(sys.argv[2])), max_frames=F)
enc.set_timing(True)
d = torch.from_numpy(pcm).cuda(); cap = enc.max_output_bytes(F)
out = torch.empty(cap, dtype=torch.uint8, device="cuda"); fb = torch.empty(F, dtype=torch.int32, device="cuda")
acc = {}
for it in range(4):
    try:
        enc.encode_device(d.data_ptr(), 1, n, out.data_ptr(), cap, fb.data_ptr())
    except engine.EngineError as ex:
        if it == 0: print("  (engine reported: %%s)" %% ex)
    if it:
        for k, v in enc.last_timing().items(): acc[k] = acc.get(k, 0) + v / 3
print("  " + " ".join("%%s=%%.3f" %% kv for kv in acc.items()), "total=%%.3f" %% sum(acc.values()))
''' % ROOT

minutes = sys.argv[1] if len(sys.argv) > 1 else "20"
level = sys.argv[2] if len(sys.argv) > 2 else "8"
for mask in sys.argv[3:] or ["0"]:
    print("FLACAMD_DEBUG=%s" % mask, flush=True)
    env = dict(os.environ, FLACAMD_DEBUG=mask)
    r = subprocess.run([sys.executable, "-c", CHILD, minutes, level], env=env, timeout=150)
    if r.returncode != 0:
        print("  child failed rc=%d" % r.returncode)
        sys.exit(1)
