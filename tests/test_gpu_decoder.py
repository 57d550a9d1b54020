"""GPU suite for the decoder kernels and the on-device MD5, through the C ABI.
Bar: decoded PCM bit-exact; digests equal to hashlib's."""
import hashlib

import numpy as np
import pytest

import oracle_bindings as ob
from golden_util import golden_names, load_golden

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", golden_names())
def test_decoder_matches_golden(gpu_engine, name):
    """Frames produced by the REFERENCE encoder decode to the original PCM."""
    g = load_golden(name)
    m = g["meta"]
    blob = b"".join(g["frames"])
    bs = max(max(ob.oracle_encode(ob.make_params(m["channels"], m["bps"], m["rate"], **{k: (v if k != "apodization" else [tuple(a) for a in v]) for k, v in m["params"].items()})[0], g["pcm"])["samples"]), 16)
    dec = gpu_engine.Decoder(m["channels"], m["bps"], bs, max_frames=len(g["frames"]) + 64, max_bytes=len(blob) + 64)
    pcm, info = dec.decode_host(blob, len(g["pcm"]) + 16)
    dec.close()
    assert info["frames"] == len(g["frames"]) and info["bytes"] == len(blob)
    assert np.array_equal(pcm, g["pcm"])


def test_decoder_int16_output(gpu_engine):
    g = load_golden("st16-level8")
    blob = b"".join(g["frames"])
    dec = gpu_engine.Decoder(2, 16, 4096, max_frames=64, max_bytes=len(blob))
    pcm, _ = dec.decode_host(blob, len(g["pcm"]), dtype=np.int16)
    dec.close()
    assert pcm.dtype == np.int16 and np.array_equal(pcm.astype(np.int32), g["pcm"])


def test_decoder_rejects_corruption(gpu_engine):
    g = load_golden("st16-level8")
    blob = bytearray(b"".join(g["frames"]))
    blob[len(g["frames"][0]) + 100] ^= 0x04   # inside the second frame: its CRC-16 fails, the chain is broken
    dec = gpu_engine.Decoder(2, 16, 4096, max_frames=64, max_bytes=len(blob))
    with pytest.raises(gpu_engine.EngineError) as ei:
        dec.decode_host(bytes(blob), len(g["pcm"]) + 16)
    dec.close()
    assert ei.value.code == -8


def test_decoder_handles_false_sync_codes(gpu_engine):
    """White noise coded verbatim is full of byte pairs that look like sync codes; every one that
    also passes the header checks is decoded speculatively and must be rejected by its CRC-16."""
    from flac_amd import synth
    pcm = synth.white_noise(60000, 2, 16, seed=5)
    fp, _ = ob.make_params(2, 16, 44100, level=8)
    o = ob.oracle_encode(fp, pcm)
    blob = b"".join(o["frames"])
    dec = gpu_engine.Decoder(2, 16, 4096, max_frames=len(o["frames"]) + 256, max_bytes=len(blob))
    out, info = dec.decode_host(blob, len(pcm))
    dec.close()
    assert info["frames"] == len(o["frames"]) and np.array_equal(out, pcm)


@pytest.mark.parametrize("flag", [0x2, 0x4, 0x8, 0x6])
def test_alternative_decoder_arrangements_give_the_same_samples(gpu_engine, monkeypatch, flag):
    """The decode kernel's later forms switched off one by one (the restorer's 16-bit history, taken in fours or one
    sample at a time; the parser's mode-free step): the forms they replaced still run where a stream calls for them, and
    must give the same samples.  Music at -8, white noise (verbatim subframes, false sync codes) and a 24-bit stream."""
    from flac_amd import synth
    streams = []
    g = load_golden("st16-level8")
    streams.append((b"".join(g["frames"]), g["pcm"], 2, 16))
    noise = synth.white_noise(30000, 2, 16, seed=9)
    fp, _ = ob.make_params(2, 16, 44100, level=8)
    o = ob.oracle_encode(fp, noise)
    streams.append((b"".join(o["frames"]), noise, 2, 16))
    m24 = synth.multichannel_24(0.4, rate=48000, channels=1, seed=3)
    fp, _ = ob.make_params(1, 24, 48000, level=8)
    o = ob.oracle_encode(fp, m24)
    streams.append((b"".join(o["frames"]), m24, 1, 24))
    monkeypatch.setenv("FLACAMD_DECODE_DEBUG", hex(flag))
    try:
        for blob, pcm, ch, bps in streams:
            dec = gpu_engine.Decoder(ch, bps, 4096, max_frames=len(blob) // 64 + 256, max_bytes=len(blob))
            out, _ = dec.decode_host(blob, len(pcm))
            dec.close()
            assert np.array_equal(out, np.asarray(pcm).reshape(len(pcm), ch))
    finally:
        monkeypatch.delenv("FLACAMD_DECODE_DEBUG")


def test_false_candidate_with_a_long_predictor(gpu_engine):
    """Bytes behind the last frame that pass for a frame header whose first subframe claims a predictor of 32
    taps: the candidate is turned down (it runs off the end), the frames before it decode as they are, and the
    decoder does not go through the batch again on its account (stream_decoder.c:2611-2619 reads such a subframe;
    the kernel with room for its history is only taken up for frames that are accepted)."""
    lib = ob.oracle_lib()
    g = load_golden("st16-level8")
    blob = b"".join(g["frames"])
    hdr = bytearray([0xFF, 0xF8, 0xC9, 0x18, 0x00])     # 4096 samples, 44.1 kHz, two independent channels, 16 bits, frame number 0
    h = np.frombuffer(bytes(hdr), dtype=np.uint8).copy()
    fake = bytes(hdr) + bytes([lib.fo_crc8(h.ctypes.data, len(h)), 0x7E]) + bytes(range(1, 120))   # 0x7E: LPC subframe of order 32
    dec = gpu_engine.Decoder(2, 16, 4096, max_frames=64, max_bytes=len(blob) + len(fake))
    for _ in range(2):   # the second call would run on the other kernel had the first switched to it: same result either way
        pcm, info = dec.decode_host(blob + fake, len(g["pcm"]) + 4096)
        assert info["frames"] == len(g["frames"]) and np.array_equal(pcm[: len(g["pcm"])], g["pcm"])
    dec.close()


def test_decoder_escape_partitions_and_rice2(gpu_engine):
    """Hand-built frames with an escaped (raw) partition and with 5-bit Rice parameters: this
    encoder never emits escapes, the decoder must still read them (stream_decoder.c:2764-2773)."""
    import struct
    lib = ob.oracle_lib()

    class W:
        def __init__(self):
            self.bits = []

        def put(self, v, n):
            self.bits += [(v >> (n - 1 - i)) & 1 for i in range(n)]

        def bytes(self):
            b = self.bits + [0] * (-len(self.bits) % 8)
            return bytes(int("".join(map(str, b[i:i + 8])), 2) for i in range(0, len(b), 8))

    n = 16
    samples = [3, -2, 100, -100, 7, 0, -1, 1, 50, -50, 2, -2, 9, -9, 1000, -1000]
    frames = b""
    for fno, (method, esc) in enumerate([(0, True), (1, True), (1, False)]):
        w = W()
        w.put(0x3FFE, 14); w.put(0, 2); w.put(6, 4); w.put(9, 4); w.put(0, 4); w.put(4, 3); w.put(0, 1)  # mono, 16 bit, blocksize byte follows
        w.put(fno, 8); w.put(n - 1, 8)
        hdr = np.frombuffer(w.bytes(), dtype=np.uint8).copy()
        w.put(lib.fo_crc8(hdr.ctypes.data, len(hdr)), 8)
        w.put(0x10, 8)                      # fixed order 0
        w.put(method, 2); w.put(0, 4)       # partition order 0
        if esc:
            w.put(31 if method else 15, 5 if method else 4); w.put(12, 5)
            for s in samples:
                w.put(s & 0xFFF, 12)
        else:
            k = 17                          # needs the 5-bit parameter field
            w.put(k, 5)
            for s in samples:
                u = (s << 1) ^ (s >> 31)
                w.put(1, (u >> k) + 1); w.put(u & ((1 << k) - 1), k)
        body = np.frombuffer(w.bytes(), dtype=np.uint8).copy()
        frames += bytes(body) + struct.pack(">H", lib.fo_crc16(body.ctypes.data, len(body)))
    want = np.array(samples * 3, dtype=np.int32).reshape(-1, 1)
    nf, ref = ob.oracle_decode(frames, 1, 16, 64)
    assert nf == 3 and np.array_equal(ref, want)
    dec = gpu_engine.Decoder(1, 16, 16, max_frames=16, max_bytes=len(frames))
    pcm, info = dec.decode_host(frames, 64)
    dec.close()
    assert info["frames"] == 3 and np.array_equal(pcm, want)


def test_full_size_encode_decode_round_trip_on_device(gpu_engine):
    """BASELINE size (one hour, -8): encode then decode entirely on the GPU; the decoded PCM
    equals the input sample for sample (compared on the device)."""
    import torch
    from flac_amd import synth
    base = synth.music_like(60, seed=1234).astype(np.int16)
    pcm = np.tile(base, (60, 1))
    n = pcm.shape[0]
    F = (n + 4095) // 4096
    dev = torch.device("cuda", 0)
    enc = gpu_engine.Encoder(gpu_engine.make_params(2, 16, 44100, level=8), max_frames=F)
    d_pcm = torch.from_numpy(pcm).to(dev)
    cap = enc.max_output_bytes(F)
    d_out = torch.empty(cap, dtype=torch.uint8, device=dev)
    d_fb = torch.empty(F, dtype=torch.int32, device=dev)
    total = enc.encode_device(d_pcm.data_ptr(), gpu_engine.PCM_S16, n, d_out.data_ptr(), cap, d_fb.data_ptr())
    enc.close()
    dec = gpu_engine.Decoder(2, 16, 4096, max_frames=F + 4096, max_bytes=total)
    d_dec = torch.zeros((n, 2), dtype=torch.int16, device=dev)
    info = dec.decode_device(d_out.data_ptr(), total, d_dec.data_ptr(), gpu_engine.PCM_S16, n)
    dec.close()
    assert info["frames"] == F and info["samples"] == n and info["bytes"] == total
    assert bool(torch.equal(d_dec, d_pcm))


def test_md5_on_device(gpu_engine):
    import torch
    rng = np.random.default_rng(11)
    dev = torch.device("cuda", 0)
    for ch, bps, dtype in ((2, 16, np.int16), (2, 16, np.int32), (1, 8, np.int32), (8, 24, np.int32), (2, 20, np.int32)):
        lens = [1, 15, 16, 17, 1000, 4096, 5000] + [int(v) for v in rng.integers(1, 3000, size=70)]
        offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
        pcm = rng.integers(-(1 << (bps - 1)), 1 << (bps - 1), size=(int(offs[-1]), ch)).astype(dtype)
        d = torch.from_numpy(pcm).to(dev)
        got = gpu_engine.md5_device(d.data_ptr(), gpu_engine.PCM_S16 if dtype == np.int16 else gpu_engine.PCM_S32, ch, bps, offs)
        nb = (bps + 7) // 8
        for i, L in enumerate(lens):
            seg = pcm[int(offs[i]):int(offs[i + 1])].astype("<i4")
            raw = seg.view(np.uint8).reshape(L, ch, 4)[:, :, :nb].tobytes()
            assert got[i] == hashlib.md5(raw).digest(), (ch, bps, i)


@pytest.mark.parametrize("world", [1, 2, 3, 8])
@pytest.mark.parametrize("kind", ["music-8", "music-5-bs1152", "mono24"])
def test_byte_range_shares_stitch_to_the_stream(gpu_engine, world, kind):
    """Decode sharding: the frame data cut into `world` byte ranges (cuts land inside frames), each share decoded
    from its own window as a rank would on its GPU, the shares stitched as rank 0 does."""
    from flac_amd import sharding, synth
    if kind == "music-8":
        pcm, ch, bps, fp = synth.music_like(1.6, seed=41)[:65000], 2, 16, ob.make_params(2, 16, 44100, level=8)[0]
    elif kind == "music-5-bs1152":
        pcm, ch, bps, fp = synth.music_like(0.8, seed=42)[:30001], 2, 16, ob.make_params(2, 16, 44100, level=5, blocksize=1152)[0]
    else:
        pcm, ch, bps, fp = synth.multichannel_24(0.2, channels=1, seed=43)[:15000], 1, 24, ob.make_params(1, 24, 96000, level=6, blocksize=576)[0]
    enc = ob.oracle_encode(fp, pcm)
    data = b"".join(enc["frames"])
    bs = max(enc["samples"])
    slack = max(len(f) for f in enc["frames"]) + 16
    dec = gpu_engine.Decoder(ch, bps, bs, max_frames=len(enc["frames"]) + 64, max_bytes=len(data) + 64)
    parts = [dec.decode_range(data, b0, b1, slack, len(pcm) + bs, len(enc["frames"]) + 64) for b0, b1 in sharding.byte_ranges(len(data), world)]
    dec.close()
    assert sum(p["frames"] for p in parts) == len(enc["frames"])
    assert np.array_equal(sharding.stitch_decoded(parts, len(data)), pcm)
    if world == 8 and kind == "music-8":
        assert any(p["frames"] == 0 for p in parts) or all(p["frames"] > 0 for p in parts)   # ranges without a frame start are legal


@pytest.mark.parametrize("name", ["st16-level8", "8ch24-8-e-p", "st16-8-bs17", "mono16-8", "fsd3", "len4097"])
def test_frames_layout_is_the_write_callbacks(gpu_engine, name):
    """FLACAMD_PCM_S32_FRAMES: every frame as channel 0's samples, then channel 1's, ... (what a decoder write callback
    is handed, stream_decoder.h:509-511), frame after frame; the same samples as the interleaved layout."""
    g = load_golden(name)
    m = g["meta"]
    blob = b"".join(g["frames"])
    sizes = ob.oracle_encode(ob.make_params(m["channels"], m["bps"], m["rate"], **{k: (v if k != "apodization" else [tuple(a) for a in v]) for k, v in m["params"].items()})[0], g["pcm"])["samples"]
    dec = gpu_engine.Decoder(m["channels"], m["bps"], max(max(sizes), 16), max_frames=len(g["frames"]) + 64, max_bytes=len(blob) + 64)
    flat, info = dec.decode_host_frames(blob, len(g["pcm"]) + 16)
    dec.close()
    assert info["frames"] == len(g["frames"]) and info["samples"] == len(g["pcm"])
    ch, at = m["channels"], 0
    for n in sizes:
        frame = flat[at * ch:(at + n) * ch].reshape(ch, n)
        assert np.array_equal(frame.T, g["pcm"][at:at + n]), (name, at)
        at += n
    assert at == len(g["pcm"])
