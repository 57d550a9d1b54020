"""GPU tests of the stream decoder object on damaged input: the same damaged bytes go through this
library's decoder and through the reference decoder (oracle/_ref/libflacref.so, built from the reference's
sources), both driven through the same ctypes wrapper, and everything the client can observe must agree:
the order of error reports and frames, sample numbers, blocksizes, the PCM (zeroed frames included), the
return value of process_until_end_of_stream, the final state and the MD5 verdict of finish()."""
import ctypes as C
import zlib

import numpy as np
import pytest

import oracle_bindings as ob
from flac_amd import stream, synth

pytestmark = pytest.mark.gpu

_ref_dec = None


def _ref_decoder_lib():
    global _ref_dec
    if _ref_dec is None:
        _ref_dec = stream.bind_decoder_api(C.CDLL(ob.REF_SO), own=False)
    return _ref_dec


_streams = {}


def _stream(kind):
    """(whole stream, offset of the first frame, frame sizes) as the reference encoder writes it"""
    if kind not in _streams:
        if kind == "st16":
            pcm = synth.music_like(1.2, seed=5)[:48000]
            _, rp = ob.make_params(2, 16, 44100, level=5, blocksize=1152)
        elif kind == "st16-8":
            pcm = synth.music_like(2.0, seed=6)[:70000]
            _, rp = ob.make_params(2, 16, 44100, level=8)
        elif kind == "mono24":
            pcm = synth.multichannel_24(0.25, channels=1, seed=9)[:20000]
            _, rp = ob.make_params(1, 24, 96000, level=6, blocksize=576)
        else:
            pcm = synth.wasted_bits(30000, channels=2, bps=16, shift=2, seed=3)
            _, rp = ob.make_params(2, 16, 44100, level=3, blocksize=2304)
        r = ob.ref_encode(rp, pcm)
        # the driver's write callback cannot seek, so the STREAMINFO it leaves is the unpatched one: put in what
        # the reference writes back into a file (frame sizes, total samples, MD5)
        bs = rp.blocksize or 4096
        head = ob.stream_head(bs, bs, r["min_framesize"], r["max_framesize"], rp.sample_rate, rp.channels, rp.bits_per_sample,
                              r["total_samples"], r["md5"], is_last=False)
        _streams[kind] = (head + r["stream"][42:], r["audio_offset"], [len(f) for f in r["frames"]])
    return _streams[kind]


def _run(lib, data, batch=None, read_chunk=None, md5=True):
    d = stream.StreamDecoder(bytes(data), seekable=False, read_chunk=read_chunk, lib=lib)
    if lib is None and batch:
        assert d.set_batch_frames(batch)
    assert d.set_md5_checking(md5)
    assert d.init() == 0
    ok = d.process_until_end_of_stream()
    state = d.get_state()
    pcm = d.pcm()
    fin = d.finish()
    out = dict(ok=ok, state=state, events=list(d.events), pcm=pcm, finish=fin)
    d.close()
    return out


def _same(data, batch=None, read_chunk=None):
    ref = _run(_ref_decoder_lib(), data)
    got = _run(None, data, batch=batch, read_chunk=read_chunk)
    assert got["events"] == ref["events"]
    assert got["ok"] == ref["ok"] and got["state"] == ref["state"]
    assert got["pcm"].shape == ref["pcm"].shape and np.array_equal(got["pcm"], ref["pcm"])
    assert got["finish"] == ref["finish"]
    return ref


def _frame_start(offsets, k):
    return offsets[k]


def _offsets(audio, sizes):
    o = [audio]
    for s in sizes:
        o.append(o[-1] + s)
    return o


@pytest.mark.parametrize("kind", ["st16", "st16-8", "mono24", "wasted"])
def test_undamaged_streams_agree(gpu_engine, kind):
    data, _, _ = _stream(kind)
    ref = _same(data, batch=3)
    assert ref["ok"] and ref["finish"] and all(e[0] == "frame" for e in ref["events"])


@pytest.mark.parametrize("kind,batch", [("st16", None), ("st16", 2), ("st16-8", 1), ("mono24", 4), ("wasted", None)])
def test_bit_flip_near_the_end_of_a_frame_gives_a_zeroed_frame(gpu_engine, kind, batch):
    data, audio, sizes = _stream(kind)
    off = _offsets(audio, sizes)
    bad = bytearray(data)
    bad[off[3] - 5] ^= 0x01        # low bits of the last residuals of frame 2: the frame keeps its length
    ref = _same(bad, batch=batch)
    assert ("error", stream.ERR_FRAME_CRC_MISMATCH) in ref["events"]


@pytest.mark.parametrize("kind,where,mask,batch", [
    ("st16", 0, 0x80, None),      # subframe type byte of the first subframe: reserved bit
    ("st16", 0, 0x40, 3),         # another subframe type
    ("st16", 1, 0xff, None),
    ("st16", 20, 0x10, 2),        # warm-up samples / coefficients
    ("st16", 60, 0x04, None),     # early residuals: token lengths change, the frame ends elsewhere
    ("st16-8", 35, 0x20, 2),
    ("st16-8", 200, 0xff, None),
    ("mono24", 12, 0x08, 1),
    ("wasted", 3, 0x01, None),
])
def test_damage_inside_a_frame_body(gpu_engine, kind, where, mask, batch):
    data, audio, sizes = _stream(kind)
    off = _offsets(audio, sizes)
    bad = bytearray(data)
    hdr = 6 if kind != "st16-8" else 6        # fixed-blocksize headers of these streams: 4 + 1 number byte + CRC-8
    bad[off[4] + hdr + where] ^= mask
    _same(bad, batch=batch)


@pytest.mark.parametrize("kind,byte,mask,batch", [
    ("st16", 1, 0x02, None), ("st16", 2, 0xf0, 2), ("st16", 2, 0x0f, None), ("st16", 3, 0xf0, None), ("st16", 3, 0x01, 3),
    ("st16", 4, 0x80, None), ("st16", 4, 0xff, None), ("st16", 5, 0x01, 1), ("st16", 0, 0x01, None), ("st16", 2, 0xff, None),
    ("mono24", 3, 0x0e, None), ("st16-8", 2, 0x60, 2),
])
def test_damage_in_a_frame_header(gpu_engine, kind, byte, mask, batch):
    data, audio, sizes = _stream(kind)
    off = _offsets(audio, sizes)
    bad = bytearray(data)
    bad[off[5] + byte] ^= mask
    _same(bad, batch=batch)


@pytest.mark.parametrize("seed,batch", [(1, None), (2, 2), (3, 1), (4, None), (5, 5)])
def test_junk_between_frames(gpu_engine, seed, batch):
    data, audio, sizes = _stream("st16")
    off = _offsets(audio, sizes)
    rng = np.random.default_rng(seed)
    junk = bytes(rng.integers(0, 256, int(rng.integers(1, 400)), dtype=np.uint8))
    if seed == 4:
        junk = b"\xff\xf8\xc9\x18" + junk + b"\xff\xff\xf8"      # things that look like sync codes
    k = 2 + seed
    bad = data[:off[k]] + junk + data[off[k]:]
    _same(bad, batch=batch)


@pytest.mark.parametrize("cut_from,cut_len,batch", [(10, 100, None), (0, 50, 2), (300, 700, None), (5, 3000, 3)])
def test_missing_bytes(gpu_engine, cut_from, cut_len, batch):
    data, audio, sizes = _stream("st16")
    off = _offsets(audio, sizes)
    a = off[6] + cut_from
    bad = data[:a] + data[a + cut_len:]
    _same(bad, batch=batch)


@pytest.mark.parametrize("fill,batch", [(0x00, None), (0xff, 2), (0xaa, None)])
def test_overwritten_run(gpu_engine, fill, batch):
    data, audio, sizes = _stream("st16-8")
    off = _offsets(audio, sizes)
    bad = bytearray(data)
    a = off[3] + 100
    bad[a:a + 300] = bytes([fill]) * 300
    _same(bad, batch=batch)


@pytest.mark.parametrize("keep,batch", [(1, None), (7, None), (100, 2), (-1, None), (-3, 1)])
def test_stream_that_ends_inside_a_frame(gpu_engine, keep, batch):
    data, audio, sizes = _stream("st16")
    off = _offsets(audio, sizes)
    end = off[9] + keep if keep > 0 else off[9] + keep
    ref = _same(data[:end], batch=batch)
    assert ref["state"] == stream.DEC_END_OF_STREAM


@pytest.mark.parametrize("tail", [b"\x00" * 37, b"TAG" + b"x" * 125, b"\xff", b"\xff\xff", b"\xff\xf8"])
def test_bytes_after_the_last_frame_of_a_stream_of_unknown_length(gpu_engine, tail):
    # with total_samples in STREAMINFO the decoder stops by itself; clear it so that the tail is looked at
    data, audio, sizes = _stream("st16")
    bad = bytearray(data)
    bad[4 + 4 + 13] &= 0xf0
    bad[4 + 4 + 14:4 + 4 + 18] = b"\x00\x00\x00\x00"
    _same(bytes(bad) + tail, batch=4)


@pytest.mark.parametrize("seed", range(12))
def test_random_damage(gpu_engine, seed):
    kind = ["st16", "st16-8", "mono24", "wasted"][seed % 4]
    data, audio, sizes = _stream(kind)
    rng = np.random.default_rng(zlib.crc32(b"damage%d" % seed))
    bad = bytearray(data)
    for _ in range(int(rng.integers(1, 6))):
        p = int(rng.integers(audio, len(bad)))
        bad[p] ^= int(rng.integers(1, 256))
    _same(bad, batch=[None, 1, 2, 7][seed % 4], read_chunk=[None, 1000, None][seed % 3])


# ---- frames whose header is valid but disagrees with STREAMINFO -------------------------------------------------

def _run_frames(lib, data, batch=None):
    d = stream.StreamDecoder(bytes(data), seekable=False, lib=lib)
    if lib is None and batch:
        assert d.set_batch_frames(batch)
    assert d.set_md5_checking(True)
    assert d.init() == 0
    ok = d.process_until_end_of_stream()
    out = dict(ok=ok, state=d.get_state(), events=list(d.events), frames=[(h, p.copy()) for h, p in d.frames],
               last=(d.get_channels(), d.get_bits_per_sample(), d.get_sample_rate(), d.get_blocksize()))
    out["finish"] = d.finish()
    d.close()
    return out


def _foreign_frames(kind):
    if kind == "mono16":
        pcm = synth.music_like(0.3, seed=41)[:9000, :1]
        _, rp = ob.make_params(1, 16, 44100, level=5, blocksize=1152)
    elif kind == "st24":
        pcm = synth.multichannel_24(0.1, rate=44100, channels=2, seed=42)[:4000]
        _, rp = ob.make_params(2, 24, 44100, level=8, blocksize=1152)
    elif kind == "st8":
        pcm = (synth.music_like(0.2, seed=43)[:5000] >> 8).astype(np.int32)
        _, rp = ob.make_params(2, 8, 44100, level=3, blocksize=576)
    elif kind == "long":    # same format, blocks longer than the stream's maximum
        pcm = synth.music_like(0.5, seed=44)[:3 * 4096]
        _, rp = ob.make_params(2, 16, 44100, level=8, blocksize=4096)
    elif kind == "rate":    # same format, another sample rate in the header
        pcm = synth.music_like(0.2, seed=45)[:3 * 1152]
        _, rp = ob.make_params(2, 16, 32000, level=5, blocksize=1152)
    else:                   # 6 channels
        pcm = synth.multichannel_24(0.05, rate=44100, channels=6, seed=46)[:2000]
        _, rp = ob.make_params(6, 24, 44100, level=5, blocksize=1152)
    return ob.ref_encode(rp, pcm)["frames"]


@pytest.mark.parametrize("kind,where,count,batch", [("mono16", 5, 2, None), ("st24", 0, 1, 7), ("st8", 11, 3, None), ("long", 7, 2, 3), ("rate", 3, 2, None),
                                                    ("ch6", 9, 1, None), ("mono16", 40, 1, 2)])
def test_frames_with_parameters_of_their_own(gpu_engine, kind, where, count, batch):
    """The reference decodes a frame by its own header (stream_decoder.c:2002-2145): frames of another stream spliced in
    come out with their own channels / sample size / blocksize / sample rate, and the stream goes on after them."""
    data, audio, sizes = _stream("st16")
    offs = _offsets(audio, sizes)
    cut = offs[min(where, len(sizes) - 1)]
    foreign = b"".join(_foreign_frames(kind)[:count])
    spliced = data[:cut] + foreign + data[cut:]
    ref = _run_frames(_ref_decoder_lib(), spliced)
    got = _run_frames(None, spliced, batch=batch)
    assert got["events"] == ref["events"]
    assert got["ok"] == ref["ok"] and got["state"] == ref["state"] and got["finish"] == ref["finish"] and got["last"] == ref["last"]
    assert len(got["frames"]) == len(ref["frames"])
    shapes = set()
    for (hg, pg), (hr, pr) in zip(got["frames"], ref["frames"]):
        assert hg == hr
        assert pg.shape == pr.shape and np.array_equal(pg, pr)
        shapes.add((hg["channels"], hg["bits_per_sample"], hg["sample_rate"]))
    assert len(shapes) == 2 or kind == "long"
