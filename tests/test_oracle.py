"""CPU suite: pins the oracle (oracle/liboracle.so, this repo's restatement of the reference
frame coder) against reference-generated golden vectors and, when oracle/_ref is built,
against the real reference run live."""
import ctypes as C
import hashlib

import numpy as np
import pytest

import oracle_bindings as ob
from cases import parity_cases
from golden_util import golden_names, load_golden


@pytest.mark.parametrize("name", golden_names())
def test_oracle_matches_golden(name):
    g = load_golden(name)
    m = g["meta"]
    fp, _ = ob.make_params(m["channels"], m["bps"], m["rate"], **{k: (v if k != "apodization" else [tuple(a) for a in v]) for k, v in m["params"].items()})
    o = ob.oracle_encode(fp, g["pcm"])
    assert len(o["frames"]) == len(g["frames"])
    for i, (a, b) in enumerate(zip(o["frames"], g["frames"])):
        assert a == b, "frame %d differs" % i
    assert o["md5"] == g["md5"]
    assert min(map(len, g["frames"])) == m["min_framesize"] and max(map(len, g["frames"])) == m["max_framesize"]


@pytest.mark.parametrize("name", golden_names())
def test_oracle_decoder_round_trip_golden(name):
    g = load_golden(name)
    m = g["meta"]
    nf, pcm = ob.oracle_decode(b"".join(g["frames"]), m["channels"], m["bps"], len(g["pcm"]) + 16)
    assert nf == len(g["frames"])
    assert np.array_equal(pcm, g["pcm"])


@pytest.mark.skipif(not ob.have_ref(), reason="oracle/_ref not built")
@pytest.mark.parametrize("case", [c for c in parity_cases(small=True)], ids=lambda c: c[0])
def test_oracle_matches_live_reference(case):
    name, make, ch, bps, rate, kw = case
    pcm = make()
    fp, rp = ob.make_params(ch, bps, rate, **kw)
    r = ob.ref_encode(rp, pcm, chunk=2048 if "level" in name else 0)  # the CLI feeds 2048-sample chunks (src/flac/encode.c:65)
    o = ob.oracle_encode(fp, pcm)
    assert len(r["frames"]) == len(o["frames"])
    assert all(a == b for a, b in zip(r["frames"], o["frames"]))
    assert r["md5"] == o["md5"]
    # the reference decoder accepts the oracle's frames (with a stream head) and returns the PCM, MD5 verified
    bs = fp.blocksize or (4096 if fp.max_lpc_order else 1152)
    head = ob.stream_head(bs, bs, min(map(len, o["frames"])), max(map(len, o["frames"])), rate, ch, bps, len(pcm), o["md5"])
    rc, dec, info = ob.ref_decode(head + b"".join(o["frames"]), len(pcm) + 16, ch)
    assert rc == 0 and info[4] == 0 and info[5] == 1
    assert np.array_equal(dec, pcm)


@pytest.mark.skipif(not ob.have_ref(), reason="oracle/_ref not built")
def test_windows_match_reference():
    ref = C.CDLL(ob.REF_SO)
    names = {"bartlett": "FLAC__window_bartlett", "bartlett_hann": "FLAC__window_bartlett_hann", "blackman": "FLAC__window_blackman",
             "blackman_harris_4term_92db": "FLAC__window_blackman_harris_4term_92db_sidelobe", "connes": "FLAC__window_connes",
             "flattop": "FLAC__window_flattop", "hamming": "FLAC__window_hamming", "hann": "FLAC__window_hann",
             "kaiser_bessel": "FLAC__window_kaiser_bessel", "nuttall": "FLAC__window_nuttall", "rectangle": "FLAC__window_rectangle",
             "triangle": "FLAC__window_triangle", "welch": "FLAC__window_welch"}
    lib = ob.oracle_lib()
    for L in (16, 17, 33, 1152, 4096, 4097):
        for name, sym in names.items():
            a = np.zeros(L, np.float32)
            b = np.zeros(L, np.float32)
            getattr(ref, sym)(C.c_void_p(a.ctypes.data), C.c_int32(L))
            lib.fo_window(ob.WINDOWS.index(name), 0.0, L, b.ctypes.data)
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (name, L)
        for p in (0.0, 0.01, 0.25, 0.5, 0.9, 1.0):
            a = np.zeros(L, np.float32)
            b = np.zeros(L, np.float32)
            ref.FLAC__window_tukey(C.c_void_p(a.ctypes.data), C.c_int32(L), C.c_float(p))
            lib.fo_window(ob.WINDOWS.index("tukey"), p, L, b.ctypes.data)
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), ("tukey", p, L)
        for sd in (0.05, 0.2, 0.5):
            a = np.zeros(L, np.float32)
            b = np.zeros(L, np.float32)
            ref.FLAC__window_gauss(C.c_void_p(a.ctypes.data), C.c_int32(L), C.c_float(sd))
            lib.fo_window(ob.WINDOWS.index("gauss"), sd, L, b.ctypes.data)
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), ("gauss", sd, L)


def test_md5_matches_hashlib():
    rng = np.random.default_rng(3)
    lib = ob.oracle_lib()
    for ch, bps, n in ((2, 16, 10000), (1, 8, 777), (8, 24, 5000), (2, 20, 4097), (2, 12, 1)):
        pcm = rng.integers(-(1 << (bps - 1)), 1 << (bps - 1), size=(n, ch)).astype(np.int32)
        nb = (bps + 7) // 8
        raw = pcm.astype("<i4").view(np.uint8).reshape(n, ch, 4)[:, :, :nb].tobytes()
        d = np.zeros(16, np.uint8)
        lib.fo_md5_pcm(pcm.ctypes.data, n, ch, bps, d.ctypes.data)
        assert bytes(d) == hashlib.md5(raw).digest()


def test_crc_known_answers():
    lib = ob.oracle_lib()
    # CRC-8 poly 0x07 and CRC-16 poly 0x8005, init 0, no reflection ("123456789" check values 0xF4 and 0xFEE8)
    s = np.frombuffer(b"123456789", dtype=np.uint8).copy()
    assert lib.fo_crc8(s.ctypes.data, 9) == 0xF4
    assert lib.fo_crc16(s.ctypes.data, 9) == 0xFEE8
    # first frame header of the reference CLI's -8 output documented in the survey: FF F8 C9 98 00 -> CRC8 0x74
    h = np.array([0xFF, 0xF8, 0xC9, 0x98, 0x00], dtype=np.uint8)
    assert lib.fo_crc8(h.ctypes.data, 5) == 0x74


def test_utf8_known_answers():
    """Byte patterns at every length boundary, the values the reference's bitwriter test
    walks (src/test_libFLAC/bitwriter.c:240-554)."""
    lib = ob.oracle_lib()
    kat = {0x00: "00", 0x7F: "7f", 0x80: "c280", 0x7FF: "dfbf", 0x800: "e0a080", 0xFFFF: "efbfbf", 0x10000: "f0908080",
           0x1FFFFF: "f7bfbfbf", 0x200000: "f888808080", 0x3FFFFFF: "fbbfbfbfbf", 0x4000000: "fc8480808080", 0x7FFFFFFF: "fdbfbfbfbfbf"}
    for v, hexs in kat.items():
        out = np.zeros(8, np.uint8)
        n = C.c_uint32(0)
        lib.fo_utf8(v, out.ctypes.data, C.byref(n))
        assert bytes(out[:n.value]).hex() == hexs, hex(v)


def test_decoder_rejects_corruption():
    g = load_golden("st16-level8")
    blob = bytearray(b"".join(g["frames"]))
    blob[len(g["frames"][0]) // 2] ^= 0x10
    nf, _ = ob.oracle_decode(bytes(blob), 2, 16, len(g["pcm"]) + 16)
    assert nf < 0


def test_frames_do_not_depend_on_neighbours():
    """A block coded in the middle of a stream equals the same block coded alone with the
    same frame number (the property the GPU batching and multi-GPU sharding rest on)."""
    g = load_golden("st16-level8")
    fp, _ = ob.make_params(2, 16, 44100, level=8)
    whole = ob.oracle_encode(fp, g["pcm"])
    for k in range(len(whole["frames"]) - 1):
        part = ob.oracle_encode(fp, g["pcm"][k * 4096:(k + 1) * 4096], first_frame=k, final=False)
        assert part["frames"][0] == whole["frames"][k]
