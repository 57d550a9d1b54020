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

# whole files: WAVE on disk -> .flac on disk through the front end, against the reference tool on the same file
import struct, subprocess, tempfile  # noqa: E402
from flac_amd import files  # noqa: E402

tmp = tempfile.mkdtemp(prefix="flacamd_rate_")
wav = os.path.join(tmp, "in.wav")
data = pcm16.tobytes()
with open(wav, "wb") as f:
    f.write(b"RIFF" + struct.pack("<I", 36 + len(data)) + b"WAVEfmt " + struct.pack("<IHHIIHH", 16, 1, 2, 44100, 176400, 4, 16) + b"data" + struct.pack("<I", len(data)))
    f.write(data)
for rep in range(2):
    t0 = time.perf_counter()
    r = files.encode_file(wav, os.path.join(tmp, "gpu.flac"), level=8, batch_frames=8192)
    dt = time.perf_counter() - t0
print("%-28s %d min: %.1f ms, %.1f MSamples/s (%d bytes out)" % ("encode_file wav->flac -8", minutes, dt * 1e3, n / dt / 1e6, r["bytes_written"]))
t0 = time.perf_counter()
r = files.decode_file(os.path.join(tmp, "gpu.flac"), os.path.join(tmp, "gpu.wav"))
dt = time.perf_counter() - t0
print("%-28s %d min: %.1f ms, %.1f MSamples/s" % ("decode_file flac->wav", minutes, dt * 1e3, n / dt / 1e6))
assert open(os.path.join(tmp, "gpu.wav"), "rb").read()[44:] == data
ref = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "flac")
if os.path.exists(ref) and minutes <= 10:
    t0 = time.perf_counter()
    subprocess.run([ref, "--silent", "-f", "-8", "-o", os.path.join(tmp, "ref.flac"), wav], check=True)
    dt = time.perf_counter() - t0
    same = open(os.path.join(tmp, "ref.flac"), "rb").read() == open(os.path.join(tmp, "gpu.flac"), "rb").read()
    print("%-28s %d min: %.1f ms, %.1f MSamples/s (identical file: %s)" % ("reference flac -8", minutes, dt * 1e3, n / dt / 1e6, same))
    t0 = time.perf_counter()
    subprocess.run([ref, "--silent", "-f", "-d", "-o", os.path.join(tmp, "ref.wav"), os.path.join(tmp, "ref.flac")], check=True)
    dt = time.perf_counter() - t0
    print("%-28s %d min: %.1f ms, %.1f MSamples/s" % ("reference flac -d", minutes, dt * 1e3, n / dt / 1e6))
for fn in os.listdir(tmp):
    os.remove(os.path.join(tmp, fn))
os.rmdir(tmp)
