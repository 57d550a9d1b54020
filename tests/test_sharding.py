"""The N>1 path on CPU: two gloo ranks each encode their frame range of one stream (the
oracle stands in for the GPU engine, which is what a rank would call on its own GPU) and
rank 0 stitches; the result must equal the single-process stream."""
import multiprocessing as mp
import os
import socket
import sys

import numpy as np
import pytest

import oracle_bindings as ob
from flac_amd import sharding, synth

HERE = os.path.dirname(os.path.abspath(__file__))


def test_frame_ranges_cover_and_align():
    for nframes in (0, 1, 2, 7, 646, 38760):
        for world in (1, 2, 3, 8):
            for align in (1, 4, 11):
                r = sharding.frame_ranges(nframes, world, align)
                assert len(r) == world and r[0][0] == 0 and r[-1][1] == nframes
                for (a, b), (c, d) in zip(r, r[1:]):
                    assert b == c and a <= b
                assert all(a % align == 0 or a == nframes for a, _ in r)
    assert sharding.frame_ranges(38760, 8) == [(i * 4845, (i + 1) * 4845) for i in range(8)]


def test_loose_period_matches_reference_formula():
    # (unsigned)((float)rate * 0.4 / (float)blocksize + 0.5), at least 1 (stream_encoder.c:905-908)
    assert sharding.loose_mid_side_period(44100, 1152) == 15
    assert sharding.loose_mid_side_period(44100, 4096) == 4
    assert sharding.loose_mid_side_period(8000, 16384) == 1
    assert sharding.loose_mid_side_period(96000, 4608) == 8


def test_stitch_scan_and_extremes():
    s = sharding.stitch([[3, 5], [], [2]], [b"aaabbbbb", b"", b"cc"])
    assert s["data"] == b"aaabbbbbcc" and list(s["offsets"]) == [0, 3, 8, 10] and (s["min_framesize"], s["max_framesize"]) == (2, 5)
    with pytest.raises(ValueError):
        sharding.stitch([[3]], [b"ab"])


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rank_main(rank, world, port, level, seconds, nsamples, q):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pcm = synth.music_like(seconds, seed=99)[:nsamples]
        fp, _ = ob.make_params(2, 16, 44100, level=level)
        bs = 4096 if fp.max_lpc_order else 1152
        align = sharding.loose_mid_side_period(44100, bs) if fp.loose_mid_side else 1

        def encode_range(first_frame, s0, s1, final):
            return ob.oracle_encode(fp, pcm[s0:s1], first_frame=first_frame, final=final)["frames"]

        got = sharding.encode_sharded(encode_range, len(pcm), bs, align=align)
        if rank == 0:
            whole = ob.oracle_encode(fp, pcm)["frames"]
            sizes = [len(f) for f in whole]
            ok = (got["data"] == b"".join(whole) and list(got["sizes"]) == sizes
                  and (got["min_framesize"], got["max_framesize"]) == (min(sizes), max(sizes)))
            q.put(("ok" if ok else "mismatch", len(whole)))
        else:
            assert got is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("level,seconds,nsamples", [(8, 1.5, 60000), (4, 2.0, 80000), (1, 0.6, 25000), (5, 0.1, 3000)])
def test_two_rank_encode_equals_single_stream(level, seconds, nsamples):
    """world_size 2 over gloo; level 4 / 1 = loose mid/side (ranges aligned to the decision
    period); the last case has fewer frames than ranks (rank 1 gets an empty range)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_main, args=(r, 2, port, level, seconds, nsamples, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
    for p in procs:
        if p.is_alive():
            p.kill()
            pytest.fail("rank hung")
        assert p.exitcode == 0
    status, nframes = q.get(timeout=5)
    assert status == "ok" and nframes >= 1


# ---- decode: byte ranges ---------------------------------------------------------------------------

def _oracle_decode_range(data, channels, bps, start, end):
    """What a rank does on its GPU (flac_amd.engine.Decoder.decode_range), with the oracle's serial frame decoder:
    the first position at or after `start` from which a frame decodes with both CRCs right opens the share; it runs
    to the first frame that starts at or after `end`."""
    buf = np.frombuffer(data, dtype=np.uint8)
    p = start
    while p + 1 < min(end, len(data)):
        if buf[p] == 0xFF and (buf[p + 1] & 0xFE) == 0xF8:
            offs = np.full(4096, 0xFFFFFFFF, dtype=np.uint32)
            lib = ob.oracle_lib()
            pcm = np.zeros((200000, channels), dtype=np.int32)
            ns = ob.C.c_uint64(0)
            tail = buf[p:]
            lib.fo_decode_frames(tail.ctypes.data, len(tail), channels, bps, pcm.ctypes.data, len(pcm), ob.C.byref(ns), offs.ctypes.data, len(offs))
            good = offs[offs != 0xFFFFFFFF].astype(np.int64)
            if len(good):
                mine = good[good + p < end]
                # bytes of exactly the frames of this share: up to the start of the first frame beyond it, or to where decoding stopped
                beyond = good[good + p >= end]
                stop = int(beyond[0]) if len(beyond) else len(tail)      # the test streams are intact: decoding ran to the end
                nf, part = ob.oracle_decode(bytes(tail[:stop]), channels, bps, 200000)
                assert nf == len(mine)
                return dict(first_pos=p, end_pos=p + stop, frames=int(nf), pcm=part)
        p += 1
    e = min(end, len(data))
    return dict(first_pos=e, end_pos=e, frames=0, pcm=np.zeros((0, channels), np.int32))


def test_byte_ranges_and_stitch_checks():
    assert sharding.byte_ranges(10, 3) == [(0, 4), (4, 8), (8, 10)]
    assert sharding.byte_ranges(2, 4) == [(0, 1), (1, 2), (2, 2), (2, 2)]
    a = dict(first_pos=0, end_pos=7, frames=2, pcm=np.ones((3, 2), np.int32))
    b = dict(first_pos=7, end_pos=7, frames=0, pcm=np.zeros((0, 2), np.int32))
    c = dict(first_pos=7, end_pos=12, frames=1, pcm=np.full((2, 2), 5, np.int32))
    assert sharding.stitch_decoded([a, b, c], 12).shape == (5, 2)
    with pytest.raises(ValueError):
        sharding.stitch_decoded([a, dict(c, first_pos=8)], 12)      # a gap between two shares
    with pytest.raises(ValueError):
        sharding.stitch_decoded([a], 12)                            # frames missing at the end


def _decode_rank_main(rank, world, port, level, nsamples, q):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pcm = synth.music_like(nsamples / 44100.0 + 0.1, seed=98)[:nsamples]
        fp, _ = ob.make_params(2, 16, 44100, level=level)
        data = b"".join(ob.oracle_encode(fp, pcm)["frames"])
        got = sharding.decode_sharded(lambda b0, b1: _oracle_decode_range(data, 2, 16, b0, b1), len(data), 2)
        if rank == 0:
            q.put("ok" if np.array_equal(got, pcm) else "mismatch")
        else:
            assert got is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("level,nsamples", [(8, 60000), (2, 30000), (5, 3000)])
def test_two_rank_decode_equals_the_pcm(level, nsamples):
    """world_size 2 over gloo, the stream cut in the middle of a frame; the last case is a single frame (rank 1's
    range holds no frame start)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_decode_rank_main, args=(r, 2, port, level, nsamples, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
    for p in procs:
        if p.is_alive():
            p.kill()
            pytest.fail("rank hung")
        assert p.exitcode == 0
    assert q.get(timeout=5) == "ok"
