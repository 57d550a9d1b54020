"""GPU tests of the whole-file front end (include/flac_amd_files.h): .flac files byte for byte the
ones the reference's flac tool writes from the same WAVE / AIFF / raw input with the same options,
and decoded WAVE / AIFF files byte for byte the ones `flac -d` writes."""
import os
import struct
import subprocess

import numpy as np
import pytest

import oracle_bindings as ob
from flac_amd import files, synth

pytestmark = pytest.mark.gpu


def _pack(pcm, width, big=False, unsigned=False, shift=0):
    x = (pcm.astype(np.int64) << shift)
    if unsigned:
        x = x + (1 << (8 * width - 1))
    x = x & ((1 << (8 * width)) - 1)
    b = np.zeros(x.shape + (width,), np.uint8)
    for i in range(width):
        b[..., (width - 1 - i) if big else i] = (x >> (8 * i)) & 255
    return b.tobytes()


def wave_bytes(pcm, rate, bps, container=None, extensible=False, mask=0, extra_chunks=b"", pad_fmt=False):
    ch = pcm.shape[1]
    container = container or ((bps + 7) // 8) * 8
    width = container // 8
    data = _pack(pcm, width, unsigned=container <= 8, shift=container - bps)
    fmt = struct.pack("<HHIIHH", 65534 if extensible else 1, ch, rate, rate * ch * width, ch * width, container)
    if extensible:
        fmt += struct.pack("<HHI", 22, bps, mask) + b"\x01\x00\x00\x00\x00\x00\x10\x00\x80\x00\x00\xaa\x00\x38\x9b\x71"
    elif pad_fmt:
        fmt += struct.pack("<H", 0)
    body = b"WAVE" + b"fmt " + struct.pack("<I", len(fmt)) + fmt + extra_chunks + b"data" + struct.pack("<I", len(data)) + data + (b"\0" if len(data) & 1 else b"")
    return b"RIFF" + struct.pack("<I", len(body)) + body


def aiff_bytes(pcm, rate, bps, offset=0):
    ch = pcm.shape[1]
    width = (bps + 7) // 8
    data = _pack(pcm, width, big=True, shift=8 * width - bps)
    sh = 0
    while (rate >> (31 - sh)) == 0:
        sh += 1
    ext = struct.pack(">HII", 63 - (sh + 32) + 0x3FFF, (rate << sh) & 0xFFFFFFFF, 0)
    comm = struct.pack(">HIH", ch, len(pcm), bps) + ext
    ssnd = struct.pack(">II", offset, 0) + b"\xAA" * offset + data
    body = b"AIFF" + b"ANNO" + struct.pack(">I", 3) + b"hi!\0" + b"COMM" + struct.pack(">I", 18) + comm + b"SSND" + struct.pack(">I", len(ssnd)) + ssnd + (b"\0" if len(ssnd) & 1 else b"")
    return b"FORM" + struct.pack(">I", len(body)) + body


def _ref(args):
    if not os.path.exists(ob.REF_FLAC):
        pytest.skip("oracle/_ref/flac not built")
    subprocess.run([ob.REF_FLAC, "--silent"] + [str(a) for a in args], check=True, timeout=180)


CASES = {
    "wav16-default": dict(kind="wav", seconds=23.0, ch=2, bps=16, rate=44100, level=5, tool=[], kw={}),
    "wav16-8-verify-tags": dict(kind="wav", seconds=3.0, ch=2, bps=16, rate=44100, level=8, tool=["-V", "-T", "ARTIST=a", "-T", "TITLE=b c", "--padding=100", "-S", "1s", "-S", "X"],
                                kw=dict(verify=True, tags=["ARTIST=a", "TITLE=b c"], padding=100, seekpoints="1s;X;")),
    "wav8-mono": dict(kind="wav", seconds=1.5, ch=1, bps=8, rate=22050, level=3, tool=["--no-padding"], kw=dict(padding=0)),
    "wav24-junk-chunks": dict(kind="wav", seconds=2.0, ch=2, bps=24, rate=96000, level=6, tool=["-S", "5x", "-b", "2048"], kw=dict(seekpoints="5x;", blocksize=2048),
                              extra=b"LIST" + struct.pack("<I", 5) + b"abcde\0" + b"junk" + struct.pack("<I", 4) + b"1234", pad_fmt=True),
    "wavext-6ch-24": dict(kind="wav", seconds=1.0, ch=6, bps=24, rate=48000, level=5, tool=["-S-"], kw=dict(seekpoints=""), extensible=True, mask=0x3f),
    "wavext-20in24": dict(kind="wav", seconds=1.0, ch=2, bps=20, rate=48000, level=5, tool=["-T", "WAVEFORMATEXTENSIBLE_CHANNEL_MASK=0x9999", "-T", "A=b"],
                          kw=dict(tags=["WAVEFORMATEXTENSIBLE_CHANNEL_MASK=0x9999", "A=b"]), extensible=True, mask=0, container=24),
    "wav16-lax-b65535": dict(kind="wav", seconds=5.0, ch=2, bps=16, rate=44100, level=8, tool=["--lax", "-b", "65535"], kw=dict(lax=True, blocksize=65535)),
    "wav24-lax-b32768-V": dict(kind="wav", seconds=1.5, ch=2, bps=24, rate=96000, level=5, tool=["--lax", "-b", "32768", "-V"], kw=dict(lax=True, blocksize=32768, verify=True)),
    "aiff16": dict(kind="aiff", seconds=2.5, ch=2, bps=16, rate=44100, level=7, tool=[], kw={}),
    "aiff12-mono-offset": dict(kind="aiff", seconds=1.0, ch=1, bps=12, rate=32000, level=2, tool=["--lax"], kw=dict(lax=True), offset=6),
    "raw16-be": dict(kind="raw", seconds=1.2, ch=2, bps=16, rate=48000, level=4, tool=["--force-raw-format", "--endian=big", "--sign=signed", "--channels=2", "--bps=16", "--sample-rate=48000", "-S", "100", "-S", "40000"],
                     kw=dict(input_format=files.RAW, raw=dict(channels=2, bps=16, sample_rate=48000, big_endian=True), seekpoints="100;40000;")),
    "raw8-unsigned": dict(kind="raw", seconds=0.7, ch=1, bps=8, rate=8000, level=0, tool=["--force-raw-format", "--endian=little", "--sign=unsigned", "--channels=1", "--bps=8", "--sample-rate=8000"],
                          kw=dict(input_format=files.RAW, raw=dict(channels=1, bps=8, sample_rate=8000, unsigned=True))),
}


def _input(c):
    n = int(c["seconds"] * c["rate"])
    base = synth.music_like(c["seconds"] + 0.1, rate=c["rate"], seed=len(c["tool"]) + c["ch"])[:n]
    cols = [base[:, i % 2].astype(np.int64) * (1 if i < 2 else (i + 1)) // (1 if i < 2 else 7) for i in range(c["ch"])]
    pcm = np.stack(cols, axis=1)
    pcm = (pcm << max(0, c["bps"] - 16)) >> max(0, 16 - c["bps"])
    if c["kind"] == "wav":
        return pcm, wave_bytes(pcm, c["rate"], c["bps"], container=c.get("container"), extensible=c.get("extensible", False), mask=c.get("mask", 0),
                               extra_chunks=c.get("extra", b""), pad_fmt=c.get("pad_fmt", False)), "in.wav"
    if c["kind"] == "aiff":
        return pcm, aiff_bytes(pcm, c["rate"], c["bps"], offset=c.get("offset", 0)), "in.aiff"
    width = c["bps"] // 8
    return pcm, _pack(pcm, width, big=c["kw"]["raw"].get("big_endian", False), unsigned=c["kw"]["raw"].get("unsigned", False)), "in.raw"


@pytest.mark.parametrize("name", sorted(CASES))
def test_encoded_file_equals_the_reference_tools(gpu_engine, tmp_path, name):
    c = CASES[name]
    pcm, blob, fname = _input(c)
    src = tmp_path / fname
    src.write_bytes(blob)
    want = tmp_path / "ref.flac"
    _ref(["-%d" % c["level"]] + c["tool"] + ["-o", want, src])
    got = tmp_path / "gpu.flac"
    r = files.encode_file(src, got, level=c["level"], batch_frames=16, **c["kw"])
    assert got.read_bytes() == want.read_bytes()
    assert r["samples"] == len(pcm) and r["channels"] == c["ch"] and r["bits_per_sample"] == c["bps"] and r["bytes_written"] == want.stat().st_size


@pytest.mark.parametrize("name", ["wav16-default", "wav8-mono", "wavext-6ch-24", "wavext-20in24", "aiff12-mono-offset", "wav24-junk-chunks", "wav16-lax-b65535"])
@pytest.mark.parametrize("fmt", ["wave", "aiff"])
def test_decoded_file_equals_the_reference_tools(gpu_engine, tmp_path, name, fmt):
    c = CASES[name]
    pcm, blob, fname = _input(c)
    src = tmp_path / fname
    src.write_bytes(blob)
    flac = tmp_path / "x.flac"
    files.encode_file(src, flac, level=c["level"], **c["kw"])
    want = tmp_path / ("ref." + fmt)
    _ref(["-d"] + (["--force-aiff-format"] if fmt == "aiff" else []) + ["-o", want, flac])
    got = tmp_path / ("gpu." + fmt)
    r = files.decode_file(flac, got, files.AIFF if fmt == "aiff" else files.WAVE)
    assert got.read_bytes() == want.read_bytes()
    assert r["samples"] == len(pcm) and r["md5_ok"]


def test_unknown_length_stream_gets_its_header_fixed_up(gpu_engine, tmp_path):
    """A stream whose STREAMINFO has no total (written to an unseekable sink): sizes are patched afterwards (decode.c:941-962)."""
    from flac_amd import stream
    n = 30000
    pcm = synth.music_like(1.0, seed=3)[:n]
    e = stream.MemoryEncoder(seekable=False)
    assert e.set_compression_level(5) and e.init() == stream.INIT_OK and e.process_interleaved(pcm) and e.finish()
    flac = tmp_path / "u.flac"
    flac.write_bytes(bytes(e.buf))
    want, got = tmp_path / "ref.wav", tmp_path / "gpu.wav"
    _ref(["-d", "-o", want, flac])
    r = files.decode_file(flac, got)
    assert got.read_bytes() == want.read_bytes() and r["samples"] == n


def test_raw_output_and_errors(gpu_engine, tmp_path):
    pcm = synth.music_like(0.5, seed=4)[:20000]
    src = tmp_path / "a.wav"
    src.write_bytes(wave_bytes(pcm, 44100, 16))
    flac = tmp_path / "a.flac"
    files.encode_file(src, flac, level=1)
    raw = tmp_path / "a.raw"
    files.decode_file(flac, raw, files.RAW)
    assert raw.read_bytes() == pcm.astype("<i2").tobytes()
    bad = bytearray(flac.read_bytes())
    bad[-100] ^= 0x55
    (tmp_path / "bad.flac").write_bytes(bytes(bad))
    with pytest.raises(files.FileError) as ei:
        files.decode_file(tmp_path / "bad.flac", tmp_path / "bad.wav")
    assert ei.value.code == -8
    with pytest.raises(files.FileError):
        files.encode_file(src, tmp_path / "b.flac", tags=["no equals sign"])


RANGES = [
    ("wav16-default", "44100", None), ("wav16-default", "0:02.5", "0:10"), ("wav16-default", None, "-3000"), ("wav16-default", "1000", "+0:01.25"),
    ("wav16-default", "0:21.75", "-0"), ("aiff16", "4097", "-0:00.5"), ("raw16-be", "12345", "40000"), ("wav24-junk-chunks", "0:01", None), ("wav8-mono", None, "5"),
]


@pytest.mark.parametrize("name,skip,until", RANGES)
def test_encoded_sample_range_equals_the_reference_tools(gpu_engine, tmp_path, name, skip, until):
    """flac --skip / --until while encoding: the same range, seek table, padding and totals as the tool's file."""
    c = CASES[name]
    pcm, blob, fname = _input(c)
    src = tmp_path / fname
    src.write_bytes(blob)
    want, got = tmp_path / "ref.flac", tmp_path / "gpu.flac"
    rng = (["--skip=" + skip] if skip else []) + (["--until=" + until] if until else [])
    _ref(["-%d" % c["level"]] + c["tool"] + rng + ["-o", want, src])
    r = files.encode_file(src, got, level=c["level"], batch_frames=16, skip=skip, until=until, **c["kw"])
    assert got.read_bytes() == want.read_bytes()
    assert 0 < r["samples"] < len(pcm)


@pytest.mark.parametrize("name,skip,until", RANGES)
@pytest.mark.parametrize("fmt", ["wave", "aiff"])
def test_decoded_sample_range_equals_the_reference_tools(gpu_engine, tmp_path, name, skip, until, fmt):
    """flac -d --skip / --until: the header promises the range, decoding starts inside a frame and stops inside one."""
    c = CASES[name]
    pcm, blob, fname = _input(c)
    src = tmp_path / fname
    src.write_bytes(blob)
    flac = tmp_path / "x.flac"
    files.encode_file(src, flac, level=c["level"], **c["kw"])
    want, got = tmp_path / ("ref." + fmt), tmp_path / ("gpu." + fmt)
    rng = (["--skip=" + skip] if skip else []) + (["--until=" + until] if until else [])
    _ref(["-d"] + (["--force-aiff-format"] if fmt == "aiff" else []) + rng + ["-o", want, flac])
    r = files.decode_file(flac, got, files.AIFF if fmt == "aiff" else files.WAVE, skip=skip, until=until)
    assert got.read_bytes() == want.read_bytes()
    assert 0 < r["samples"] < len(pcm)


@pytest.mark.parametrize("skip,until", [("+5", None), ("abc", None), (None, "1:xx"), ("100", "50"), (None, "999999999"), ("999999999", None), (None, "-999999999"), ("0:05", "+0")])
def test_sample_range_errors(gpu_engine, tmp_path, skip, until):
    """What the tool refuses is refused: relative --skip, malformed values, --until before the skip point / outside the input."""
    pcm = synth.music_like(0.5, seed=4)[:20000]
    src = tmp_path / "a.wav"
    src.write_bytes(wave_bytes(pcm, 44100, 16))
    if (skip, until) != ("0:05", "+0"):
        with pytest.raises(files.FileError):
            files.encode_file(src, tmp_path / "a.flac", level=1, skip=skip, until=until)
    flac = tmp_path / "whole.flac"
    files.encode_file(src, flac, level=1)
    with pytest.raises(files.FileError):
        files.decode_file(flac, tmp_path / "a.out.wav", skip=skip, until=until)


_W64 = {k: (k.encode() + b"\xF3\xAC\xD3\x11\xD1\x8C\x00\xC0\x4F\x8E\xDB\x8A") for k in ("wave", "fmt ", "data", "junk")}
_W64["riff"] = b"riff\x2E\x91\xCF\x11\xD6\xA5\x28\xDB\x04\xC1\x00\x00"


def _fmt_body(pcm, rate, bps, extensible, mask):
    ch, width = pcm.shape[1], (bps + 7) // 8
    fmt = struct.pack("<HHIIHH", 65534 if extensible else 1, ch, rate, rate * ch * width, ch * width, width * 8)
    if extensible:
        fmt += struct.pack("<HHI", 22, bps, mask) + b"\x01\x00\x00\x00\x00\x00\x10\x00\x80\x00\x00\xaa\x00\x38\x9b\x71"
    return fmt


def rf64_bytes(pcm, rate, bps, extensible=False, mask=0, extra=b"", data_size_in_chunk=False):
    width = (bps + 7) // 8
    data = _pack(pcm, width, unsigned=bps <= 8, shift=8 * width - bps)
    fmt = _fmt_body(pcm, rate, bps, extensible, mask)
    tail = b"fmt " + struct.pack("<I", len(fmt)) + fmt + extra + b"data" + struct.pack("<I", len(data) if data_size_in_chunk else 0xFFFFFFFF) + data + (b"\0" if len(data) & 1 else b"")
    ds64 = b"ds64" + struct.pack("<IQQQI", 28, 4 + 36 + len(tail), len(data), len(pcm), 0)
    return b"RF64" + struct.pack("<I", 0xFFFFFFFF) + b"WAVE" + ds64 + tail


def wave64_bytes(pcm, rate, bps, extensible=False, mask=0, junk=b""):
    width = (bps + 7) // 8
    data = _pack(pcm, width, unsigned=bps <= 8, shift=8 * width - bps)
    fmt = _fmt_body(pcm, rate, bps, extensible, mask)

    def chunk(guid, body):
        return guid + struct.pack("<Q", 24 + len(body)) + body + b"\0" * (-len(body) % 8)
    body = _W64["wave"] + chunk(_W64["fmt "], fmt) + (chunk(_W64["junk"], junk) if junk else b"") + chunk(_W64["data"], data)
    return _W64["riff"] + struct.pack("<Q", 24 + len(body)) + body


WIDE_CASES = {
    "rf64-16": dict(make=lambda p: rf64_bytes(p, 44100, 16), ch=2, bps=16, rate=44100, seconds=2.0, name="in.rf64"),
    "rf64-24-6ch-extra": dict(make=lambda p: rf64_bytes(p, 48000, 24, extensible=True, mask=0x3f, extra=b"LIST" + struct.pack("<I", 3) + b"abc\0", data_size_in_chunk=True),
                              ch=6, bps=24, rate=48000, seconds=0.8, name="in.rf64"),
    "w64-16": dict(make=lambda p: wave64_bytes(p, 44100, 16, junk=b"12345"), ch=2, bps=16, rate=44100, seconds=2.0, name="in.w64"),
    "w64-8-mono-odd": dict(make=lambda p: wave64_bytes(p[:-3], 22050, 8), ch=1, bps=8, rate=22050, seconds=0.7, name="in.w64", drop=3),
    "w64-20in24": dict(make=lambda p: wave64_bytes(p, 96000, 20, extensible=True, mask=0x3), ch=2, bps=20, rate=96000, seconds=0.5, name="in.w64"),
}


def _wide_pcm(c):
    n = int(c["seconds"] * c["rate"])
    base = synth.music_like(c["seconds"] + 0.1, rate=c["rate"], seed=c["ch"] + c["bps"])[:n]
    pcm = np.stack([base[:, i % 2].astype(np.int64) for i in range(c["ch"])], axis=1)
    return (pcm << max(0, c["bps"] - 16)) >> max(0, 16 - c["bps"])


@pytest.mark.parametrize("name", sorted(WIDE_CASES))
def test_rf64_and_wave64_input_equals_the_reference_tools(gpu_engine, tmp_path, name):
    c = WIDE_CASES[name]
    pcm = _wide_pcm(c)
    src = tmp_path / c["name"]
    src.write_bytes(c["make"](pcm))
    want, got = tmp_path / "ref.flac", tmp_path / "gpu.flac"
    _ref(["-6", "-o", want, src])
    r = files.encode_file(src, got, level=6, batch_frames=16)
    assert got.read_bytes() == want.read_bytes()
    assert r["samples"] == len(pcm) - c.get("drop", 0) and r["bits_per_sample"] == c["bps"]


@pytest.mark.parametrize("name", ["wav16-default", "wav8-mono", "wavext-6ch-24", "wavext-20in24", "aiff12-mono-offset"])
@pytest.mark.parametrize("fmt", ["rf64", "w64"])
def test_decoded_rf64_and_wave64_equal_the_reference_tools(gpu_engine, tmp_path, name, fmt):
    """flac -d --force-rf64-format / --force-wave64-format, whole and as a sample range (odd byte counts: the pad rules differ)."""
    c = CASES[name]
    pcm, blob, fname = _input(c)
    src = tmp_path / fname
    src.write_bytes(blob)
    flac = tmp_path / "x.flac"
    files.encode_file(src, flac, level=c["level"], **c["kw"])
    for skip, until in ((None, None), ("777", "-0:00.1")):
        want, got = tmp_path / ("ref." + fmt), tmp_path / ("gpu." + fmt)
        rng = (["--skip=" + skip] if skip else []) + (["--until=" + until] if until else [])
        _ref(["-d", "-f", "--force-rf64-format" if fmt == "rf64" else "--force-wave64-format"] + rng + ["-o", want, flac])
        files.decode_file(flac, got, files.RF64 if fmt == "rf64" else files.WAVE64, skip=skip, until=until)
        assert got.read_bytes() == want.read_bytes()
        # and back in: the file just written is read like the tool reads it
        back_ref, back_gpu = tmp_path / "back_ref.flac", tmp_path / "back_gpu.flac"
        _ref(["-f", "-3", "-o", back_ref, want])
        files.encode_file(got, back_gpu, level=3)
        assert back_gpu.read_bytes() == back_ref.read_bytes()


def _png(w=4, h=3):
    import zlib
    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xffffffff)
    raw = b"".join(b"\0" + bytes([i * 20 % 256] * (w * 3)) for i in range(h))
    return b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(raw)) + chunk(b"IEND", b"")


RECODE = [
    ("plain", [], {}, [], {}),
    ("tags-kept", ["-T", "ARTIST=x", "-T", "TITLE=y z", "-S", "1s", "--padding=300"], dict(tags=["ARTIST=x", "TITLE=y z"], seekpoints="1s;", padding=300), [], {}),
    ("tags-replaced", ["-T", "ARTIST=x"], dict(tags=["ARTIST=x"]), ["-T", "ALBUM=q"], dict(tags=["ALBUM=q"])),
    ("seek-dropped", ["-S", "0.5s"], dict(seekpoints="0.5s;"), ["-S-", "--no-padding"], dict(seekpoints="", padding=0)),
    ("seek-replaced-pad", ["-S", "0.5s"], dict(seekpoints="0.5s;"), ["-S", "3x", "-P", "77"], dict(seekpoints="3x;", padding=77)),
    ("range", ["-T", "A=b", "-S", "1s"], dict(tags=["A=b"], seekpoints="1s;"), ["--skip=0:01", "--until=-5000"], dict(skip="0:01", until="-5000")),
    ("picture-cuesheet", ["PICTURE", "CUESHEET"], None, [], {}),
    ("picture-cuesheet-range", ["PICTURE", "CUESHEET"], None, ["--until=0:02"], dict(until="0:02")),
    ("mask-carried", ["6CH"], None, ["-T", "X=y"], dict(tags=["X=y"])),
]


@pytest.mark.parametrize("name,first_tool,first_kw,tool,kw", RECODE, ids=[r[0] for r in RECODE])
def test_flac_input_is_coded_again_like_the_reference_tools(gpu_engine, tmp_path, name, first_tool, first_kw, tool, kw):
    """FLAC in, FLAC out: decoded on the device and coded again; tags, seek table, padding, pictures and cue sheets kept,
    replaced or dropped as the tool does it (src/flac/encode.c:1746-2015)."""
    rate, n = 44100, 44100 * 3
    six = "6CH" in first_tool
    pcm = synth.music_like(3.1, seed=12)[:n]
    if six:
        pcm = np.stack([pcm[:, i % 2] // (i + 1) for i in range(6)], axis=1)
    src = tmp_path / "in.wav"
    src.write_bytes(wave_bytes(pcm, rate, 16, extensible=six, mask=0x3f if six else 0))
    first = tmp_path / "first.flac"
    extra = []
    if "PICTURE" in first_tool:
        (tmp_path / "p.png").write_bytes(_png())
        (tmp_path / "c.cue").write_text('FILE "in.wav" WAVE\n  TRACK 01 AUDIO\n    INDEX 01 00:00:00\n  TRACK 02 AUDIO\n    INDEX 01 00:01:37\n')
        extra = ["--picture=3|image/png|cover||" + str(tmp_path / "p.png"), "--cuesheet=" + str(tmp_path / "c.cue")]
    if first_kw is None:
        _ref(["-2"] + extra + ["-o", first, src])
    else:
        _ref(["-2"] + first_tool + ["-o", first, src])
        mine = tmp_path / "first_gpu.flac"
        files.encode_file(src, mine, level=2, **first_kw)
        assert mine.read_bytes() == first.read_bytes()
    want, got = tmp_path / "ref.flac", tmp_path / "gpu.flac"
    _ref(["-7"] + tool + ["-o", want, first])
    r = files.encode_file(first, got, level=7, batch_frames=8, **kw)
    assert got.read_bytes() == want.read_bytes()
    assert r["channels"] == pcm.shape[1] and 0 < r["samples"] <= n
