"""GPU tests of the libFLAC-style stream encoder object (include/flac_amd_stream.h): the
whole stream a client receives through the callbacks - head, frames, callback arguments,
back-patched STREAMINFO - against the reference's output for the same calls."""
import os
import subprocess

import numpy as np
import pytest

import oracle_bindings as ob
from golden_util import load_golden
from flac_amd import stream, synth

pytestmark = pytest.mark.gpu

VENDOR = b"reference libFLAC 1.2.1 20070917"
VC_BLOCK = bytes([0x84, 0, 0, 8 + len(VENDOR)]) + len(VENDOR).to_bytes(4, "little") + VENDOR + bytes(4)


def _configure(e, ch, bps, rate, kw):
    """Apply a parity case's settings through the public setters, the way the flac tool does."""
    assert e.set_streamable_subset(0)
    assert e.set_channels(ch) and e.set_bits_per_sample(bps) and e.set_sample_rate(rate)
    kw = dict(kw)
    assert e.set_compression_level(kw.pop("level", 8))
    if "blocksize" in kw:
        assert e.set_blocksize(kw.pop("blocksize"))
    apod = kw.pop("apodization", None)
    if apod:
        assert e.set_apodization(";".join("%s(%g)" % (n, p) if n in ("tukey", "gauss") else n for n, p in apod))
    names = dict(do_mid_side="set_do_mid_side_stereo", loose_mid_side="set_loose_mid_side_stereo", max_lpc_order="set_max_lpc_order",
                 qlp_coeff_precision="set_qlp_coeff_precision", do_qlp_coeff_prec_search="set_do_qlp_coeff_prec_search",
                 do_exhaustive_model_search="set_do_exhaustive_model_search", min_residual_partition_order="set_min_residual_partition_order",
                 max_residual_partition_order="set_max_residual_partition_order", disable_constant="disable_constant_subframes",
                 disable_fixed="disable_fixed_subframes", disable_verbatim="disable_verbatim_subframes")
    for k, v in kw.items():
        assert getattr(e, names[k])(v), k


def _feed(e, pcm, chunk):
    for i in range(0, len(pcm), chunk):
        assert e.process_interleaved(pcm[i:i + chunk])


@pytest.mark.parametrize("name,batch,chunk", [
    ("st16-level5", 1024, 4096), ("st16-level8", 2, 1000), ("st16-8-bs17", 7, 333), ("mono16-8", 1, 4096), ("8ch24-8", 3, 5000),
    ("multi-apod", 1024, 12345), ("loose-4", 5, 4608), ("len5", 4, 1), ("len4097", 1, 4097), ("wasted", 1024, 1 << 20),
    ("disable-all", 2, 777), ("rate12345", 1024, 100),
])
def test_seekable_stream_matches_reference(gpu_engine, name, batch, chunk):
    g = load_golden(name)
    m = g["meta"]
    e = stream.MemoryEncoder(seekable=True)
    _configure(e, m["channels"], m["bps"], m["rate"], m["params"])
    assert e.set_batch_frames(batch)
    assert e.init() == stream.INIT_OK and e.get_state() == stream.OK
    assert not e.set_channels(1)  # setters are refused once initialised
    bs = e.get_blocksize()
    _feed(e, g["pcm"], chunk)
    assert e.finish() and e.get_state() == stream.UNINITIALIZED
    sizes = [len(f) for f in g["frames"]]
    head = ob.stream_head(bs, bs, min(sizes), max(sizes), m["rate"], m["channels"], m["bps"], len(g["pcm"]), g["md5"], is_last=False)
    want = head + VC_BLOCK + b"".join(g["frames"])
    assert bytes(e.buf) == want
    # callbacks: 3 head writes, one per frame (stream_encoder.c:2431-2518), 3 back-patches (:2521-2620)
    n = len(g["pcm"])
    frames = [(sizes[i], min(bs, n - i * bs), i) for i in range(len(sizes))]
    assert e.writes == [(4, 0, 0), (38, 0, 0), (len(VC_BLOCK), 0, 0)] + frames + [(16, 0, 0), (5, 0, 0), (6, 0, 0)]
    si = e.streaminfo
    assert si["md5sum"] == g["md5"] and si["total_samples"] == n and (si["min_framesize"], si["max_framesize"]) == (min(sizes), max(sizes))
    assert (si["min_blocksize"], si["max_blocksize"], si["channels"], si["bits_per_sample"], si["sample_rate"]) == (bs, bs, m["channels"], m["bps"], m["rate"])
    e.close()


@pytest.mark.parametrize("seed,level,n,batch", [(1, 8, 50000, 4), (2, 5, 4096 * 3, 2), (3, 0, 1152 * 5 + 1, 5), (4, 4, 100000, 3)])
def test_unseekable_stream_matches_live_reference(gpu_engine, seed, level, n, batch):
    """No seek callback: STREAMINFO keeps its zero frame sizes/MD5 and the estimate (:1310-1377)."""
    if not ob.have_ref():
        pytest.skip("oracle/_ref not built")
    pcm = synth.music_like(n / 44100.0 + 0.01, seed=seed)[:n]
    _, rp = ob.make_params(2, 16, 44100, level=level)
    ref = ob.ref_encode(rp, pcm, chunk=997)
    e = stream.MemoryEncoder(seekable=False)
    _configure(e, 2, 16, 44100, dict(level=level))
    assert e.set_batch_frames(batch)
    assert e.init() == stream.INIT_OK
    half = n // 2
    assert e.process([pcm[:half, 0], pcm[:half, 1]])       # planar entry (:2004-2060)
    assert e.process_interleaved(pcm[half:])
    assert e.finish()
    assert bytes(e.buf) == ref["stream"]
    assert e.streaminfo["md5sum"] == ref["md5"] and e.streaminfo["total_samples"] == n
    assert (e.streaminfo["min_framesize"], e.streaminfo["max_framesize"]) == (ref["min_framesize"], ref["max_framesize"])
    e.close()


@pytest.mark.parametrize("bs,level,bps,n", [(65535, 8, 16, 150000), (32768, 5, 16, 100000), (20000, 6, 24, 50000), (16385, 8, 16, 40000)])
def test_long_blocks_outside_the_subset(gpu_engine, bs, level, bps, n):
    """--lax block sizes above 16384 (the format allows 65535): stream equal to the live reference's, verify mode on,
    and back through the decoder object."""
    if not ob.have_ref():
        pytest.skip("oracle/_ref not built")
    pcm = synth.music_like(n / 44100.0 + 0.01, seed=bs)[:n] if bps == 16 else synth.sine(n, 2, 24, f2=3000.0)
    rate = 44100 if bps == 16 else 96000
    _, rp = ob.make_params(2, bps, rate, level=level, blocksize=bs)
    ref = ob.ref_encode(rp, pcm, chunk=4099)
    e = stream.MemoryEncoder(seekable=False)
    _configure(e, 2, bps, rate, dict(level=level, blocksize=bs))
    assert e.set_verify(1) and e.set_batch_frames(2)
    assert e.init() == stream.INIT_OK
    _feed(e, pcm, 30000)
    assert e.finish()
    assert bytes(e.buf) == ref["stream"]
    e.close()
    d = stream.StreamDecoder(ref["stream"], seekable=False)
    assert d.set_batch_frames(2) and d.init() == 0
    assert d.process_until_end_of_stream() and d.errors == []
    assert np.array_equal(d.pcm(), pcm)
    assert d.finish()
    d.close()


def test_frames_beyond_the_verbatim_size(gpu_engine):
    """Full-scale 24-bit noise: a narrow partition sum of the reference wraps (stream_encoder.c:3808), the estimate of a
    fixed subframe falls far short and the reference writes a frame of twice the verbatim size.  The same stream has to
    come out of the encoder object (second pack pass with large slots, output buffer grown), verified, and decode again."""
    if not ob.have_ref():
        pytest.skip("oracle/_ref not built")
    pcm = np.random.default_rng(3).integers(-7300000, 7300000, (3000, 2)).astype(np.int32)
    _, rp = ob.make_params(2, 24, 48000, level=8, blocksize=1000, max_residual_partition_order=4)
    ref = ob.ref_encode(rp, pcm)
    assert max(len(f) for f in ref["frames"]) > 1000 * 2 * 3 * 3 // 2       # the case is what it says
    e = stream.MemoryEncoder(seekable=False)
    _configure(e, 2, 24, 48000, dict(level=8, blocksize=1000, max_residual_partition_order=4))
    assert e.set_verify(1) and e.set_batch_frames(2)
    assert e.init() == stream.INIT_OK
    _feed(e, pcm, 1234)
    assert e.finish()
    assert bytes(e.buf) == ref["stream"]
    e.close()
    d = stream.StreamDecoder(ref["stream"], seekable=False)
    assert d.init() == 0 and d.process_until_end_of_stream() and d.errors == []
    assert np.array_equal(d.pcm(), pcm)
    d.close()


def test_file_encoder_matches_reference_tool(gpu_engine, tmp_path):
    """init_file + progress callback; byte-for-byte the file the reference's flac tool writes."""
    if not os.path.exists(ob.REF_FLAC):
        pytest.skip("oracle/_ref/flac not built")
    n = 44100 * 3 + 123
    pcm = synth.music_like(3.1, seed=77)[:n]
    raw = tmp_path / "in.raw"
    pcm.astype("<i2").tofile(raw)
    want = tmp_path / "ref.flac"
    subprocess.run([ob.REF_FLAC, "--silent", "--force-raw-format", "--endian=little", "--sign=signed", "--channels=2", "--bps=16",
                    "--sample-rate=44100", "--no-padding", "--no-seektable", "-8", "-o", str(want), str(raw)], check=True, timeout=120)
    e = stream.FileEncoder()
    assert e.set_compression_level(8) and e.set_total_samples_estimate(n) and e.set_batch_frames(6)
    out = tmp_path / "gpu.flac"
    assert e.init(str(out)) == stream.INIT_OK
    _feed(e, pcm, 10000)
    assert e.finish()
    assert out.read_bytes() == want.read_bytes()
    nframes = (n + 4095) // 4096
    # the head writes already count as "frame 0 written" (stream_encoder.c:2511 applies to every write), so
    # the reference's callback reports 2 for the first frame and k+1 for frame k afterwards
    assert [p[2] for p in e.progress] == [2] + list(range(2, nframes + 1))
    assert e.progress[-1][1] == n and e.progress[-1][0] == out.stat().st_size and e.progress[-1][3] == nframes
    e.close()


def test_write_error_stops_the_encoder(gpu_engine):
    class Failing(stream.MemoryEncoder):
        def write_callback(self, data, samples, current_frame):
            if samples and current_frame == 2:
                return 1
            return super().write_callback(data, samples, current_frame)

    e = Failing(seekable=False)
    assert e.set_compression_level(3) and e.set_batch_frames(2)
    assert e.init() == stream.INIT_OK
    pcm = synth.music_like(1.0, seed=5)[:4096 * 6]
    # batches are in flight while later samples arrive, so the refused write is reported by the call during which
    # the batch comes back: this process call, a later one, or finish() at the latest - never lost
    ok = e.process_interleaved(pcm)
    for _ in range(200):
        if not ok:
            break
        ok = e.process_interleaved(pcm[:4096 * 2])
    assert not ok
    assert e.get_state() == stream.CLIENT_ERROR        # write_frame_ (:2513-2516)
    assert [w for w in e.writes if w[1]] == [(w[0], 4096, i) for i, w in enumerate([w for w in e.writes if w[1]])] and len([w for w in e.writes if w[1]]) == 2
    assert not e.process_interleaved(pcm[:10])
    assert not e.finish() and e.get_state() == stream.CLIENT_ERROR
    e.close()
    # and when nothing more is handed over, finish() reports it
    e = Failing(seekable=False)
    assert e.set_compression_level(3) and e.set_batch_frames(2)
    assert e.init() == stream.INIT_OK
    e.process_interleaved(pcm)
    assert not e.finish() and e.get_state() == stream.CLIENT_ERROR
    e.close()


def test_encoder_object_is_reusable_after_finish(gpu_engine):
    g = load_golden("st16-level3")
    m = g["meta"]
    e = stream.MemoryEncoder(seekable=True)
    outs = []
    for _ in range(2):
        _configure(e, m["channels"], m["bps"], m["rate"], m["params"])
        e.buf, e.pos, e.writes = bytearray(), 0, []
        assert e.init() == stream.INIT_OK
        assert e.init() == stream.INIT_ALREADY_INITIALIZED
        _feed(e, g["pcm"], 3000)
        assert e.finish()
        outs.append(bytes(e.buf))
    assert outs[0] == outs[1] and outs[0].endswith(b"".join(g["frames"]))
    e.close()
