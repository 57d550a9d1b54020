"""GPU tests of the pipelined host path: batches in flight over one or several devices must give the
stream the reference gives, whatever the batch size, depth and number of devices (two entries naming
device 0 stand in for two GPUs on a one-GPU box), through the engine-level pipe and through the
libFLAC-style object."""
import numpy as np
import pytest

import oracle_bindings as ob
from golden_util import load_golden
from flac_amd import stream, synth
from test_gpu_stream import _configure, _feed, VC_BLOCK

pytestmark = pytest.mark.gpu


def _split(buf, sizes):
    out, off = [], 0
    for s in sizes:
        out.append(bytes(buf[off:off + int(s)]))
        off += int(s)
    assert off == len(buf)
    return out


@pytest.mark.parametrize("name,batch,devices,depth", [
    ("st16-level8", 3, (0,), 2), ("st16-level8", 2, (0, 0), 2), ("st16-level8", 1, (0, 0, 0), 3), ("loose-4", 2, (0, 0), 2), ("loose-4", 11, (0,), 2),
    ("8ch24-8-e-p", 1, (0, 0), 2), ("st16-8-bs17", 5, (0, 0), 4), ("len5", 1, (0, 0), 2), ("wasted", 1, (0, 0), 2), ("st16-level1", 3, (0, 0, 0, 0), 2),
])
def test_pipe_over_devices_matches_reference_frames(gpu_engine, name, batch, devices, depth):
    g = load_golden(name)
    m = g["meta"]
    p = gpu_engine.make_params(m["channels"], m["bps"], m["rate"], **m["params"])
    pipe = gpu_engine.Pipe(p, batch, devices=devices, depth=depth)
    pcm = g["pcm"]
    for arr in ((pcm, pcm.astype(np.int16)) if m["bps"] <= 16 else (pcm,)):
        buf, sizes = pipe.encode_host(arr)
        assert _split(buf, sizes) == g["frames"]
    pipe.close()


def test_pipe_long_stream_many_batches_matches_oracle(gpu_engine):
    """Many more batches than slots, sizes that leave a short last batch and a short last block."""
    pcm = synth.music_like(12.0, seed=77).astype(np.int16)[: 4096 * 120 + 777]
    fp, _ = ob.make_params(2, 16, 44100, level=8)
    want = ob.oracle_encode(fp, pcm.astype(np.int32))["frames"]
    p = gpu_engine.make_params(2, 16, 44100, level=8)
    for batch, devices, depth in ((7, (0,), 2), (16, (0, 0), 3), (50, (0, 0, 0), 2), (200, (0,), 2)):
        pipe = gpu_engine.Pipe(p, batch, devices=devices, depth=depth)
        buf, sizes = pipe.encode_host(pcm)
        assert _split(buf, sizes) == want, (batch, devices, depth)
        buf, sizes = pipe.encode_host(pcm[:4096 * 9])     # the pipe is reusable, here for a run of whole blocks
        assert _split(buf, sizes) == want[:9]
        pipe.close()


@pytest.mark.parametrize("name,batch,devices,chunk,planar", [
    ("st16-level8", 2, (0, 0), 1000, False), ("loose-4", 3, (0, 0, 0), 4608, False), ("8ch24-8-e-p", 1, (0, 0), 5000, True),
    ("st16-level5", 4, (0, 0), 4096, True), ("mono16-8", 1, (0, 0), 333, True), ("sine24-st", 2, (0, 0), 10000, False),
])
def test_stream_object_over_devices_matches_reference(gpu_engine, name, batch, devices, chunk, planar):
    g = load_golden(name)
    m = g["meta"]
    e = stream.MemoryEncoder(seekable=True)
    _configure(e, m["channels"], m["bps"], m["rate"], m["params"])
    assert e.set_batch_frames(batch) and e.set_devices(list(devices)) and e.set_batches_in_flight(2)
    assert e.init() == stream.INIT_OK
    bs = e.get_blocksize()
    pcm = g["pcm"]
    if planar:
        for i in range(0, len(pcm), chunk):
            assert e.process([pcm[i:i + chunk, c] for c in range(m["channels"])])
    else:
        _feed(e, pcm, chunk)
    assert e.finish()
    sizes = [len(f) for f in g["frames"]]
    head = ob.stream_head(bs, bs, min(sizes), max(sizes), m["rate"], m["channels"], m["bps"], len(pcm), g["md5"], is_last=False)
    assert bytes(e.buf) == head + VC_BLOCK + b"".join(g["frames"])
    n = len(pcm)
    assert e.writes[3:-3] == [(sizes[i], min(bs, n - i * bs), i) for i in range(len(sizes))]
    assert e.streaminfo["md5sum"] == g["md5"]
    e.close()


def test_stream_object_md5_thread_and_slot_reuse(gpu_engine):
    """A stream of many small batches fed in odd chunks: the MD5 worker, the GPUs and the filling thread
    pass the same buffers round; digest and bytes must equal the oracle's."""
    import hashlib
    pcm = synth.music_like(20.0, seed=5).astype(np.int32)
    fp, _ = ob.make_params(2, 16, 44100, level=5)
    want = ob.oracle_encode(fp, pcm)
    e = stream.MemoryEncoder(seekable=True)
    assert e.set_compression_level(5) and e.set_batch_frames(2) and e.set_devices([0, 0])
    assert e.init() == stream.INIT_OK
    rng = np.random.default_rng(1)
    i = 0
    while i < len(pcm):
        n = int(rng.integers(1, 30000))
        assert e.process_interleaved(pcm[i:i + n])
        i += n
    assert e.finish()
    assert e.streaminfo["md5sum"] == hashlib.md5(pcm.astype("<i2").tobytes()).digest()
    assert bytes(e.buf)[-sum(map(len, want["frames"])):] == b"".join(want["frames"])
    e.close()


def test_encoder_after_encoder_starts_from_a_clean_pipe(gpu_engine):
    """An encoder object whose settings equal its predecessor's takes over that one's idle pipe (pinned buffers are
    dear to make).  Whatever number of batches the predecessor left behind, the newcomer's first batch must sit where
    its MD5 thread looks for it: the digest in STREAMINFO is the digest of the samples handed over."""
    import hashlib
    pcm = synth.music_like(6.0, seed=8).astype(np.int32)
    for nbatches_before in (1, 2, 4, 5):
        a = stream.MemoryEncoder(seekable=True)
        assert a.set_compression_level(5) and a.set_batch_frames(4) and a.init() == stream.INIT_OK
        assert a.process_interleaved(pcm[:4096 * 4 * (nbatches_before - 1) + 100]) and a.finish()
        a.close()
        b = stream.MemoryEncoder(seekable=True)
        assert b.set_compression_level(5) and b.set_batch_frames(4) and b.init() == stream.INIT_OK
        assert b.process_interleaved(pcm) and b.finish()
        assert b.streaminfo["md5sum"] == hashlib.md5(pcm.astype("<i2").tobytes()).digest(), nbatches_before
        b.close()
