"""GPU suite: the HIP engine, called through the C ABI, against the oracle and the golden
vectors.  Bar: frame bytes bit-exact (integer/byte work; the float stages are kept in the
reference's evaluation order so their outputs are compared bit-for-bit as well)."""
import zlib

import numpy as np
import pytest

import oracle_bindings as ob
from cases import parity_cases
from golden_util import golden_names, load_golden

pytestmark = pytest.mark.gpu


def _engine_params(engine, ch, bps, rate, kw):
    return engine.make_params(ch, bps, rate, **kw)


def _encode(engine, ch, bps, rate, kw, pcm, **enc_kw):
    import ctypes as C
    p = _engine_params(engine, ch, bps, rate, kw)
    # exactly the frames the input makes at the resolved blocksize: a capacity bug shows, and the arenas stay small
    q = engine.Params.from_buffer_copy(p)
    bs = int(q.blocksize) if engine.lib().flacamd_params_resolve(C.byref(q)) == 0 else 16
    enc = engine.Encoder(p, max_frames=max(1, -(-len(pcm) // max(bs, 1))))
    try:
        return enc, enc.encode_host(pcm, **enc_kw)
    except Exception:
        enc.close()
        raise


@pytest.mark.parametrize("name", golden_names())
def test_engine_matches_golden(gpu_engine, name):
    g = load_golden(name)
    m = g["meta"]
    kw = {k: (v if k != "apodization" else [tuple(a) for a in v]) for k, v in m["params"].items()}
    enc, (frames, _) = _encode(gpu_engine, m["channels"], m["bps"], m["rate"], kw, g["pcm"])
    enc.close()
    assert len(frames) == len(g["frames"])
    for i, (a, b) in enumerate(zip(frames, g["frames"])):
        assert a == b, "frame %d differs from the reference's" % i


@pytest.mark.parametrize("case", parity_cases(small=False), ids=lambda c: c[0])
def test_engine_matches_oracle(gpu_engine, case):
    name, make, ch, bps, rate, kw = case
    pcm = make()
    fp, _ = ob.make_params(ch, bps, rate, **kw)
    o = ob.oracle_encode(fp, pcm, want_info=True)
    enc, (frames, _) = _encode(gpu_engine, ch, bps, rate, kw, pcm)
    try:
        assert len(frames) == len(o["frames"])
        nsig = 4 if (ch == 2 and fp.do_mid_side) else ch
        F = len(frames)
        # white box first, so that a mismatch names the stage that broke
        fe = enc.debug(1, np.uint64, 5 * F * nsig).reshape(5, F * nsig)
        # (loose mid/side: the oracle, like the reference, skips the unused signal pair of follower frames)
        for f in range(F if not fp.loose_mid_side else 0):
            for s in range(min(nsig, 4)):
                if o["infos"][f].blocksize >= 4:
                    assert list(fe[:, f * nsig + s]) == list(o["infos"][f].fixed_err[s]), "fixed error sums, frame %d signal %d" % (f, s)
        if fp.max_lpc_order and fp.num_apodizations == 1 and not fp.loose_mid_side:
            lag = fp.max_lpc_order + 1
            au = enc.debug(0, np.float32, lag * F * nsig).reshape(lag, F * nsig)
            for f in range(F):
                n = o["infos"][f].blocksize
                for s in range(min(nsig, 4)):
                    want = np.array(o["infos"][f].autoc[s][:min(lag, n)], dtype=np.float32)
                    got = au[:min(lag, n), f * nsig + s]
                    if o["infos"][f].sub[0].type >= 0 and np.any(want):
                        assert np.array_equal(want.view(np.uint32), got.view(np.uint32)), "autocorrelation, frame %d signal %d" % (f, s)
        for i, (a, b) in enumerate(zip(frames, o["frames"])):
            assert a == b, "frame %d differs" % i
        if bps <= 16:  # the native-width input layout takes its own load paths
            frames16, _ = enc.encode_host(pcm.astype(np.int16))
            assert frames16 == o["frames"], "int16 input"
    finally:
        enc.close()


def test_int16_and_int32_inputs_agree(gpu_engine):
    g = load_golden("st16-level8")
    kw = dict(level=8)
    e1, (f32, _) = _encode(gpu_engine, 2, 16, 44100, kw, g["pcm"])
    e1.close()
    e2, (f16, _) = _encode(gpu_engine, 2, 16, 44100, kw, g["pcm"].astype(np.int16))
    e2.close()
    assert f32 == f16 == g["frames"]


def test_batches_split_anywhere_give_the_same_stream(gpu_engine):
    """Encoding a stream in several batches (frame numbers and the final flag carried by the
    caller) produces the frames of a single call: what the streaming shim and the multi-GPU
    frame-range sharding rely on."""
    from flac_amd import synth
    pcm = synth.music_like(5.0, seed=77)
    for level in (8, 4):  # 4 = loose mid/side: the previous assignment crosses batch borders
        fp, _ = ob.make_params(2, 16, 44100, level=level)
        whole = ob.oracle_encode(fp, pcm)["frames"]
        p = gpu_engine.make_params(2, 16, 44100, level=level)
        enc = gpu_engine.Encoder(p, max_frames=64)
        out, frame_no, prev = [], 0, 0
        cuts = [0, 4096 * 3, 4096 * 4, 4096 * 11, len(pcm)]
        for a, b in zip(cuts[:-1], cuts[1:]):
            frames, prev = enc.encode_host(pcm[a:b], final=(b == len(pcm)), first_frame=frame_no, prev_assignment=prev)
            out += frames
            frame_no += len(frames)
        enc.close()
        assert out == whole


def test_large_frame_numbers(gpu_engine):
    """Frame numbers across the UTF-8 length boundaries of the header (bitwriter.c:784-832)."""
    g = load_golden("st16-level8")
    fp, _ = ob.make_params(2, 16, 44100, level=8)
    p = gpu_engine.make_params(2, 16, 44100, level=8)
    enc = gpu_engine.Encoder(p, max_frames=8)
    for first in (0x7E, 0x7FE, 0xFFFE, 0x1FFFFE, 0x3FFFFFE, 0x7FFFFFF0):
        want = ob.oracle_encode(fp, g["pcm"], first_frame=first)["frames"]
        got, _ = enc.encode_host(g["pcm"], first_frame=first)
        assert got == want, hex(first)
    enc.close()


def test_full_size_stream_properties(gpu_engine):
    """BASELINE configuration at full size (one hour, 38 760 frames, -8) through
    size-independent properties: every frame carries valid CRCs and decodes to its PCM block
    (checked on the whole stream by the oracle decoder), and frames sampled across the hour
    equal the oracle's coding of the same block under the same frame number."""
    import torch
    from flac_amd import synth
    base = synth.music_like(60, seed=1234).astype(np.int16)
    pcm = np.tile(base, (60, 1))
    n = pcm.shape[0]
    F = (n + 4095) // 4096
    p = gpu_engine.make_params(2, 16, 44100, level=8)
    enc = gpu_engine.Encoder(p, max_frames=F)
    dev = torch.device("cuda", 0)
    d_pcm = torch.from_numpy(pcm).to(dev)
    cap = enc.max_output_bytes(F)
    d_out = torch.empty(cap, dtype=torch.uint8, device=dev)
    d_fb = torch.empty(F, dtype=torch.int32, device=dev)
    total = enc.encode_device(d_pcm.data_ptr(), gpu_engine.PCM_S16, n, d_out.data_ptr(), cap, d_fb.data_ptr())
    sizes = d_fb.cpu().numpy().astype(np.int64)
    assert int(sizes.sum()) == total and len(sizes) == F == 38760
    blob = d_out[:total].cpu().numpy().tobytes()
    enc.close()
    offs = np.concatenate([[0], np.cumsum(sizes)])
    fp, _ = ob.make_params(2, 16, 44100, level=8)
    for f in list(range(0, F, 997)) + [F - 2, F - 1]:
        a, b = f * 4096, min((f + 1) * 4096, n)
        want = ob.oracle_encode(fp, pcm[a:b].astype(np.int32), first_frame=f, final=(f == F - 1))["frames"][0]
        assert blob[offs[f]:offs[f + 1]] == want, "frame %d" % f
    # whole-stream decode (sync, CRC-8, CRC-16 of all 38 760 frames) of the first and last ten minutes
    for lo, hi in ((0, 6460), (F - 6460, F)):
        nf, dec = ob.oracle_decode(blob[offs[lo]:offs[hi]], 2, 16, (hi - lo) * 4096)
        assert nf == hi - lo
        assert np.array_equal(dec, pcm[lo * 4096:min(hi * 4096, n)].astype(np.int32))


def test_eight_channel_24bit_exhaustive_precision_search(gpu_engine):
    """flac -8 -e -p on 24-bit / 96 kHz / 8 channels: 64-bit prediction sums, 132 LPC candidates per
    channel in several candidate groups; 4 s against the oracle frame by frame, then a longer run
    through the device decoder."""
    from flac_amd import synth
    kw = dict(level=8, do_exhaustive_model_search=1, do_qlp_coeff_prec_search=1)
    pcm = synth.multichannel_24(4.0, rate=96000, channels=8, seed=4321)
    fp, _ = ob.make_params(8, 24, 96000, **kw)
    want = ob.oracle_encode(fp, pcm)["frames"]
    enc, (frames, _) = _encode(gpu_engine, 8, 24, 96000, kw, pcm)
    enc.close()
    assert frames == want
    long_pcm = synth.multichannel_24(20.0, rate=96000, channels=8, seed=99)
    enc, (frames, _) = _encode(gpu_engine, 8, 24, 96000, kw, long_pcm)
    enc.close()
    blob = b"".join(frames)
    dec = gpu_engine.Decoder(8, 24, 4096, len(frames) + 64, len(blob) + 64)
    back, info = dec.decode_host(blob, len(long_pcm))
    dec.close()
    assert info["frames"] == len(frames) and np.array_equal(back, long_pcm)
    ratio = len(blob) / (long_pcm.size * 3)
    assert 0.3 < ratio < 0.95


def _stress_signal(kind, n, rng):
    t = np.arange(n)
    if kind == "side-boundary":      # L - R crosses the 16-bit range in some frames only
        l = (rng.integers(-200, 200, n) + 16384 + 16383 * np.sign(np.sin(t / 9000.0))).clip(-32768, 32767)
        r = (rng.integers(-200, 200, n) - 16384 * np.sign(np.sin(t / 5000.0)) - (t > n // 2) * 300).clip(-32768, 32767)
    elif kind == "side-exact":       # side signal exactly -32768 / 32767 / 32768
        l = np.where((t // 4096) % 3 == 0, -32768, np.where((t // 4096) % 3 == 1, 32767, 32767)) + 0 * t
        r = np.where((t // 4096) % 3 == 0, 0, np.where((t // 4096) % 3 == 1, 0, -1)) + (t % 7 == 0) * 0
        l = l + np.where(t % 5 == 0, -1, 0) * (l > 0); r = r + 0
    elif kind == "loud-noise-tone":  # poor predictors: large predictions, small shifts
        l = (30000 * np.sin(t * 0.9) + rng.integers(-2700, 2700, n)).astype(np.int64).clip(-32768, 32767)
        r = (rng.integers(-32768, 32767, n) * (np.sin(t / 3000.0) > 0) + 20000 * np.sin(t * 2.9)).astype(np.int64).clip(-32768, 32767)
    elif kind == "spikes":           # near silence with rare full-scale spikes: k = 0 partitions beside huge ones
        l = rng.integers(-1, 2, n) + (rng.random(n) < 0.001) * rng.integers(-32768, 32767, n)
        r = rng.integers(0, 2, n) + (rng.random(n) < 0.0005) * 32767
    elif kind == "dc-steps":
        l = 12345 + 0 * t + (t // 3000) * 257 % 4000
        r = -30000 + (t // 5000) * 1111 % 9000
    elif kind == "alternating":      # +-full scale every sample, and a slow one
        l = np.where(t % 2 == 0, 32767, -32768)
        r = np.where((t // 37) % 2 == 0, 32767, -32768)
    elif kind == "wasted-mix":       # different wasted bits per channel and per frame
        l = (rng.integers(-2000, 2000, n) << np.where((t // 4096) % 2 == 0, 3, 0)).clip(-32768, 32767)
        r = (rng.integers(-500, 500, n) << 5).clip(-32768, 32767)
    else:                            # smooth: high orders win, large shifts
        l = (25000 * np.sin(t * 0.011) + 6000 * np.sin(t * 0.37 + 1)).astype(np.int64)
        r = (25000 * np.sin(t * 0.0113 + 0.3) - 6000 * np.sin(t * 0.21)).astype(np.int64)
    return np.stack([np.asarray(l, np.int64), np.asarray(r, np.int64)], axis=1).clip(-32768, 32767).astype(np.int32)


@pytest.mark.parametrize("kind", ["side-boundary", "side-exact", "loud-noise-tone", "spikes", "dc-steps", "alternating", "wasted-mix", "smooth"])
@pytest.mark.parametrize("level,bs", [(8, 4096), (5, 4096), (7, 2048), (3, 1024)])
def test_stress_signals_on_the_register_resident_path(gpu_engine, kind, level, bs):
    """Signals aimed at the shortcuts of the stereo 16-bit kernels: the side signal at the edge of 16 bits, predictions far
    from the samples, near-silent partitions next to loud ones, wasted bits that differ by channel and frame."""
    rng = np.random.default_rng(zlib.crc32(kind.encode()) + level)
    pcm = _stress_signal(kind, bs * 6 + 100, rng)
    kw = dict(level=level, blocksize=bs)
    fp, _ = ob.make_params(2, 16, 44100, **kw)
    o = ob.oracle_encode(fp, pcm)
    enc, (frames, _) = _encode(gpu_engine, 2, 16, 44100, kw, pcm)
    try:
        assert frames == o["frames"]
        frames16, _ = enc.encode_host(pcm.astype(np.int16))
        assert frames16 == o["frames"], "int16 input"
    finally:
        enc.close()


@pytest.mark.parametrize("flag", [0x80000, 0x40000, 0xC0000, 0x100000, 0x20000000])
def test_alternative_stage_arrangements_give_the_same_frames(gpu_engine, monkeypatch, flag):
    """The register-resident kernels take shortcuts that other configurations cannot: the pack kernel writes frames
    at their final place (look-back over the frame sizes) instead of into slots for the gather kernel, the search
    kernel takes the fixed predictors' partition sums from the preparation pass instead of computing them, and the
    preparation kernel for full-length frames takes the frame totals from those partition sums.  The development
    switches that turn each of them off must not change a byte.  Nor must the last switch, which sends every third
    frame down the path of a frame whose place was not known in time: into its slot and from there to its place by
    the pass that follows the pack kernel."""
    from flac_amd import synth
    pcm = synth.music_like(3.0, seed=91).astype(np.int16)
    for level in (8, 5, 2):
        fp, _ = ob.make_params(2, 16, 44100, level=level)
        want = ob.oracle_encode(fp, pcm.astype(np.int32))["frames"]
        e, (frames, _) = _encode(gpu_engine, 2, 16, 44100, dict(level=level), pcm)
        e.close()
        monkeypatch.setenv("FLACAMD_DEBUG", hex(flag))
        try:
            e, (alt, _) = _encode(gpu_engine, 2, 16, 44100, dict(level=level), pcm)
            e.close()
        finally:
            monkeypatch.delenv("FLACAMD_DEBUG")
        assert frames == want and alt == want


@pytest.mark.parametrize("flag", [0x40000000, 0x10000000])
def test_general_kernels_stand_in_for_the_special_ones(gpu_engine, monkeypatch, flag):
    """Independent channels of 32-bit words have a preparation kernel that reads a frame once for all channels, and
    one-channel streams take the register-resident autocorrelation kernel with a lane per frame; with either switched
    off the general kernels do the same work and the frames stay those of the reference encoder."""
    from flac_amd import synth
    cases = [(synth.multichannel_24(0.3, rate=96000, channels=6, seed=17), 6, 24, 96000, dict(level=8)),
             (synth.multichannel_24(0.25, rate=48000, channels=1, seed=18), 1, 24, 48000, dict(level=5)),
             (synth.music_like(0.6, seed=19)[:, :1].astype(np.int32), 1, 16, 44100, dict(level=8))]
    for pcm, ch, bps, rate, kw in cases:
        fp, _ = ob.make_params(ch, bps, rate, **kw)
        want = ob.oracle_encode(fp, np.asarray(pcm, dtype=np.int32))["frames"]
        e, (frames, _) = _encode(gpu_engine, ch, bps, rate, kw, pcm)
        e.close()
        monkeypatch.setenv("FLACAMD_DEBUG", hex(flag))
        try:
            e, (alt, _) = _encode(gpu_engine, ch, bps, rate, kw, pcm)
            e.close()
        finally:
            monkeypatch.delenv("FLACAMD_DEBUG")
        assert frames == want and alt == want


def test_estimate_spot_check_totals(gpu_engine, monkeypatch):
    """The development switch 0x400000 has the pack kernel compare each chosen subframe's estimated size with the bits
    it packs (the reference's SPOTCHECK_ESTIMATE, stream_encoder.c:3444-3473).  Constant subframes are estimated
    exactly; Rice-coded ones by the reference's approximation (stream_encoder.c:3862-3876), which must stay within a
    small fraction of the packed size.  The frames themselves do not change, and an encoder created without the
    switch refuses the query."""
    from flac_amd import synth
    music = synth.music_like(3.0, seed=17).astype(np.int16)[: 32 * 4096]
    silence = np.zeros((3 * 4096, 2), dtype=np.int16)
    plain, (want, _) = _encode(gpu_engine, 2, 16, 44100, dict(level=8), music)
    try:
        with pytest.raises(gpu_engine.EngineError):
            plain.spotcheck()
    finally:
        plain.close()
    monkeypatch.setenv("FLACAMD_DEBUG", "0x400000")
    try:
        e, (got, _) = _encode(gpu_engine, 2, 16, 44100, dict(level=8), music)
        try:
            t = e.spotcheck(reset=True)
            assert got == want
            assert t["subframes"] == 2 * len(got)
            payload = 8 * sum(len(f) for f in got)
            assert 0.97 * payload < t["packed_bits"] <= payload
            assert t["sum_abs_bits"] < 0.01 * t["packed_bits"] and t["max_abs_bits"] < 0.05 * t["packed_bits"] / t["subframes"]
            assert e.spotcheck()["subframes"] == 0, "reset"
            frames, _ = e.encode_host(silence)
            t = e.spotcheck()
            assert t["subframes"] == 2 * len(frames) and t["differing"] == 0
        finally:
            e.close()
    finally:
        monkeypatch.delenv("FLACAMD_DEBUG")


@pytest.mark.parametrize("channels", [2, 3])
def test_output_buffer_too_small_is_reported(gpu_engine, channels):
    """Frames that do not fit the caller's buffer are not written past its end: the call reports the capacity error
    (stereo: the pack kernel's in-place path; three channels: the slot-and-copy path)."""
    import torch
    from flac_amd import synth
    pcm = np.ascontiguousarray(np.stack([synth.music_like(1.0, seed=5 + k)[:, 0] for k in range(channels)], axis=1).astype(np.int32))
    p = gpu_engine.make_params(channels, 16, 44100, level=5)
    enc = gpu_engine.Encoder(p, max_frames=16)
    d_pcm = torch.from_numpy(pcm).cuda()
    guard = 4096
    d_out = torch.full((1000 + guard,), 0xA5, dtype=torch.uint8, device="cuda")
    d_fb = torch.zeros(16, dtype=torch.int32, device="cuda")
    with pytest.raises(gpu_engine.EngineError) as ei:
        enc.encode_device(d_pcm.data_ptr(), gpu_engine.PCM_S32, len(pcm), d_out.data_ptr(), 1000, d_fb.data_ptr())
    assert ei.value.code == -6
    assert bool((d_out[1000:] == 0xA5).all())
    # and with room the same encoder carries on
    d_big = torch.empty(pcm.nbytes + 4096, dtype=torch.uint8, device="cuda")
    total = enc.encode_device(d_pcm.data_ptr(), gpu_engine.PCM_S32, len(pcm), d_big.data_ptr(), d_big.numel(), d_fb.data_ptr())
    fp, _ = ob.make_params(channels, 16, 44100, level=5)
    want = b"".join(ob.oracle_encode(fp, pcm)["frames"])
    assert bytes(d_big[:total].cpu().numpy()) == want
    enc.close()
