"""GPU tests of the pipelined host path: batches in flight over one or several devices must give the
stream the reference gives, whatever the batch size, depth and number of devices, through the engine-level
pipe and through the libFLAC-style object.  The device lists below say how many entries a pipe gets: on a box
with several GPUs entry i names GPU i modulo the number present, on a one-GPU box every entry names device 0
(two entries naming one device stand in for two GPUs)."""
import numpy as np
import pytest

import oracle_bindings as ob
from golden_util import load_golden
from flac_amd import stream, synth
from test_gpu_stream import _configure, _feed, VC_BLOCK

pytestmark = pytest.mark.gpu


def _devices(entries):
    import torch
    n = torch.cuda.device_count()
    return tuple(i % n for i in range(len(entries))) if n > 1 else tuple(entries)


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
    pipe = gpu_engine.Pipe(p, batch, devices=_devices(devices), depth=depth)
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
        pipe = gpu_engine.Pipe(p, batch, devices=_devices(devices), depth=depth)
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
    assert e.set_batch_frames(batch) and e.set_devices(list(_devices(devices))) and e.set_batches_in_flight(2)
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
    assert e.set_compression_level(5) and e.set_batch_frames(2) and e.set_devices(list(_devices((0, 0))))
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


def test_queued_batches_equal_the_synchronous_call(gpu_engine):
    """flacamd_encode_device_queue: several batches of one stream queued back to back (nothing waited for in between),
    each into its own output, give the frames the synchronous entry point gives and the oracle's; a batch whose output
    buffer is too small is reported by the wait call; more than FLACAMD_QUEUE_DEPTH outstanding batches are refused."""
    import torch
    pcm = synth.music_like(9.0, seed=31).astype(np.int16)[: 4096 * 70 + 1234]
    fp, _ = ob.make_params(2, 16, 44100, level=8)
    want = ob.oracle_encode(fp, pcm.astype(np.int32))["frames"]
    p = gpu_engine.make_params(2, 16, 44100, level=8)
    per = 24                                   # frames per batch: 24, 24, 23 (the last with the short block)
    nframes = (len(pcm) + 4095) // 4096
    enc = gpu_engine.Encoder(p, max_frames=per)
    enc.set_timing(True)
    d = torch.from_numpy(pcm).cuda()
    cap = enc.max_output_bytes(per)
    outs, fbs, spans = [], [], []
    for f0 in range(0, nframes, per):
        f1 = min(f0 + per, nframes)
        ns = min(len(pcm), f1 * 4096) - f0 * 4096
        outs.append(torch.empty(cap, dtype=torch.uint8, device="cuda"))
        fbs.append(torch.zeros(per, dtype=torch.int32, device="cuda"))
        spans.append((f0, f1, ns))
        enc.encode_device_queue(d.data_ptr() + f0 * 4096 * 4, gpu_engine.PCM_S16, ns, outs[-1].data_ptr(), cap, fbs[-1].data_ptr(),
                                final=(f1 == nframes), first_frame=f0)
    totals = enc.wait()
    assert len(totals) == len(spans)
    assert all(v > 0 for v in enc.last_timing().values())
    got = []
    for (f0, f1, ns), out, fb, total in zip(spans, outs, fbs, totals):
        sizes = fb[: f1 - f0].cpu().numpy()
        assert int(sizes.sum()) == total
        got += _split(out[:total].cpu().numpy().tobytes(), sizes)
        # the synchronous call on the same batch
        out2 = torch.empty(cap, dtype=torch.uint8, device="cuda")
        fb2 = torch.zeros(per, dtype=torch.int32, device="cuda")
        t2 = enc.encode_device(d.data_ptr() + f0 * 4096 * 4, gpu_engine.PCM_S16, ns, out2.data_ptr(), cap, fb2.data_ptr(), final=(f1 == nframes), first_frame=f0)
        assert t2 == total and torch.equal(out2[:total], out[:total]) and torch.equal(fb2, fb)
    assert got == want
    assert enc.wait() == []                    # nothing outstanding
    # too small an output buffer: reported when waited for, with the size it takes
    small = torch.empty(1000, dtype=torch.uint8, device="cuda")
    enc.encode_device_queue(d.data_ptr(), gpu_engine.PCM_S16, per * 4096, small.data_ptr(), 1000, fbs[0].data_ptr(), final=False)
    with pytest.raises(gpu_engine.EngineError):
        enc.wait()
    # the queue is bounded
    for _ in range(16):
        enc.encode_device_queue(d.data_ptr(), gpu_engine.PCM_S16, per * 4096, outs[0].data_ptr(), cap, fbs[0].data_ptr(), final=False)
    with pytest.raises(gpu_engine.EngineError):
        enc.encode_device_queue(d.data_ptr(), gpu_engine.PCM_S16, per * 4096, outs[0].data_ptr(), cap, fbs[0].data_ptr(), final=False)
    assert len(enc.wait()) == 16
    enc.close()


def test_encoders_sharing_the_gpu_give_the_same_frames(gpu_engine):
    """Several encoders coding at once on one GPU from threads of their own: their pack kernels share the CUs, and a
    frame whose predecessors' sizes are late goes through its slot instead of waiting (fa_place_late_kernel).  Every
    pass of every encoder must give the bytes a pass alone gives."""
    import threading
    import torch
    pcm = np.tile(synth.music_like(20.0, seed=12).astype(np.int16), (6, 1))
    n = len(pcm)
    F = (n + 4095) // 4096
    p = gpu_engine.make_params(2, 16, 44100, level=8)
    d = torch.from_numpy(pcm).cuda()

    def run(enc, out, fb, stream):
        total = enc.encode_device(d.data_ptr(), gpu_engine.PCM_S16, n, out.data_ptr(), out.numel(), fb.data_ptr(), stream=stream.cuda_stream)
        return bytes(out[:total].cpu().numpy()), fb.cpu().numpy().copy()

    encs = [gpu_engine.Encoder(p, max_frames=F) for _ in range(4)]
    bufs = [(torch.empty(encs[0].max_output_bytes(F), dtype=torch.uint8, device="cuda"), torch.zeros(F, dtype=torch.int32, device="cuda"), torch.cuda.Stream()) for _ in encs]
    want = run(encs[0], *bufs[0])
    assert want[1].sum() == len(want[0])
    results = [[] for _ in encs]

    def worker(k):
        torch.cuda.set_device(0)
        for _ in range(6):
            results[k].append(run(encs[k], *bufs[k]))

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(len(encs))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for e in encs:
        e.close()
    for k in range(len(encs)):
        assert len(results[k]) == 6
        for got in results[k]:
            assert got[0] == want[0] and np.array_equal(got[1], want[1])
