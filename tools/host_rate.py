#!/usr/bin/env python3
"""PCIe-inclusive rates: the same -8 workload as bench.py but handed over in HOST buffers,
(a) through flacamd_encode_host and (b) through the libFLAC-style stream object (which adds
the int32 queue copy and the host MD5).  Prints one line per path."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flac_amd import engine, stream, synth  # noqa: E402

minutes = int(sys.argv[1]) if len(sys.argv) > 1 else 10
base = synth.music_like(60, seed=1234)
pcm32 = np.ascontiguousarray(np.tile(base, (minutes, 1)).astype(np.int32))
pcm16 = pcm32.astype(np.int16)
n = len(pcm32)
F = (n + 4095) // 4096
enc = engine.Encoder(engine.make_params(2, 16, 44100, level=8), max_frames=F)
for name, pcm in (("encode_host int16", pcm16), ("encode_host int32", pcm32)):
    enc.encode_host(pcm[:4096 * 64])
    t0 = time.perf_counter()
    frames, _ = enc.encode_host(pcm)
    dt = time.perf_counter() - t0
    print("%-28s %d min: %.1f ms, %.1f MSamples/s (%d bytes out)" % (name, minutes, dt * 1e3, n / dt / 1e6, sum(map(len, frames))))
enc.close()


class Sink(stream.StreamEncoder):
    total = 0

    def write_callback(self, data, samples, current_frame):
        self.total += len(data)
        return 0


for md5 in (1, 0):
    e = Sink()
    e.set_compression_level(8)
    e.set_do_md5(md5)
    e.set_batch_frames(F)
    assert e.init() == 0
    t0 = time.perf_counter()
    for i in range(0, n, 1 << 20):
        assert e.process_interleaved(pcm32[i:i + (1 << 20)])
    assert e.finish()
    dt = time.perf_counter() - t0
    print("%-28s %d min: %.1f ms, %.1f MSamples/s (%d bytes out)" % ("stream object md5=%d" % md5, minutes, dt * 1e3, n / dt / 1e6, e.total))
    e.close()
