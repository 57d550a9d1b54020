"""Compiled C clients of the boundary: the same client sources (examples/c/*.c, written against the libFLAC API)
are built against include/flac_amd_stream.h + libflacamd.so and against the reference's own headers and library;
run on the same input both must write the same files.  The decode client reads every field of FLAC__Frame
(header, subframes[], footer), so this also pins the struct layout a C client sees.

Two further builds close the loop from the reference's side (oracle/Makefile, made where the reference tree is
present): the same clients compiled against the reference's UNMODIFIED headers and linked with libflacamd.so
(refhdr_*: struct layouts and enum values come from the reference's headers, every `extern const` the headers
declare from this library), and the reference's own `flac` tool - its unmodified objects and its host-side
metadata editing code - linked with libflacamd.so for encoder, decoder and format symbols (flac_on_amd)."""
import os
import struct
import subprocess

import numpy as np
import pytest

from flac_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OURS = os.path.join(ROOT, "examples", "c", "bin")
REF = os.path.join(ROOT, "oracle", "_ref")


def _wav(path, pcm, rate):
    data = pcm.astype("<i2").tobytes()
    ch = pcm.shape[1]
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", 36 + len(data)) + b"WAVEfmt " + struct.pack("<IHHIIHH", 16, 1, ch, rate, rate * ch * 2, ch * 2, 16) + b"data" + struct.pack("<I", len(data)))
        f.write(data)


def _need(*paths):
    for p in paths:
        if not os.path.exists(p):
            pytest.skip("%s not built" % os.path.relpath(p, ROOT))


ENCODE_CLIENTS = [os.path.join(OURS, "encode_file"), os.path.join(REF, "refhdr_encode_file")]
DECODE_CLIENTS = [os.path.join(OURS, "decode_analyze"), os.path.join(REF, "refhdr_decode_analyze")]


@pytest.mark.parametrize("client", ENCODE_CLIENTS, ids=["own-header", "reference-headers"])
@pytest.mark.parametrize("level,seconds,ch", [(5, 3.0, 2), (8, 1.7, 2), (0, 2.0, 1), (4, 2.5, 2)])
def test_encode_client_writes_the_reference_clients_file(gpu_engine, tmp_path, level, seconds, ch, client):
    _need(client, os.path.join(REF, "client_encode_file"))
    pcm = synth.music_like(seconds, seed=31 + level)[:, :ch]
    wav = str(tmp_path / "in.wav")
    _wav(wav, pcm, 44100)
    a, b = str(tmp_path / "ours.flac"), str(tmp_path / "ref.flac")
    r1 = subprocess.run([client, wav, a, str(level)], capture_output=True, text=True, timeout=120)
    r2 = subprocess.run([os.path.join(REF, "client_encode_file"), wav, b, str(level)], capture_output=True, text=True, timeout=120)
    assert r1.returncode == 0, r1.stderr
    assert r2.returncode == 0, r2.stderr
    assert r1.stdout.replace("ours.flac", "x") == r2.stdout.replace("ref.flac", "x")
    assert open(a, "rb").read() == open(b, "rb").read()


@pytest.mark.parametrize("client", DECODE_CLIENTS, ids=["own-header", "reference-headers"])
@pytest.mark.parametrize("level,extra", [(8, []), (5, []), (0, []), (8, ["-b", "1152", "-r", "8"]), (3, ["--lax", "-b", "577"])])
def test_decode_client_reads_frames_like_the_reference_client(gpu_engine, tmp_path, level, extra, client):
    _need(client, os.path.join(REF, "client_decode_analyze"), os.path.join(REF, "flac"))
    pcm = synth.music_like(4.0, seed=70 + level)
    pcm[10000:30000] >>= 3
    pcm[10000:30000] <<= 3           # wasted bits
    pcm[40000:60000, 1] = 1234       # a constant channel
    wav = str(tmp_path / "in.wav")
    _wav(wav, pcm, 44100)
    flac = str(tmp_path / "in.flac")
    subprocess.run([os.path.join(REF, "flac"), "--silent", "-f", "-%d" % level] + extra + ["-o", flac, wav], check=True, timeout=120)
    outs = []
    for exe, tag in ((client, "ours"), (os.path.join(REF, "client_decode_analyze"), "ref")):
        w, a = str(tmp_path / (tag + ".wav")), str(tmp_path / (tag + ".ana"))
        r = subprocess.run([exe, flac, w, a], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, (tag, r.stderr)
        outs.append((open(w, "rb").read(), open(a).read()))
    assert outs[0][0] == outs[1][0] == open(wav, "rb").read()
    assert outs[0][1] == outs[1][1]
    assert "type=LPC" in outs[0][1] or level == 0


# ---- the reference's command line tool over this library ----
TOOL_REF = os.path.join(REF, "flac")
TOOL_AMD = os.path.join(REF, "flac_on_amd")


def _tool(exe, args, cwd):
    r = subprocess.run([exe] + args, capture_output=True, text=True, timeout=300, cwd=cwd)
    return r.returncode, r.stdout, r.stderr


def _wav24(path, pcm, rate):
    ch = pcm.shape[1]
    raw = pcm.astype("<i4").reshape(-1, 1).view(np.uint8).reshape(-1, 4)[:, :3].tobytes()
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", 36 + len(raw)) + b"WAVEfmt " + struct.pack("<IHHIIHH", 16, 1, ch, rate, rate * ch * 3, ch * 3, 24) + b"data" + struct.pack("<I", len(raw)))
        f.write(raw)


@pytest.mark.parametrize("args", [
    ["-8"], ["-5"], ["-0"], ["-8", "--verify"], ["-8", "-e", "-p"], ["-4", "-b", "1152"], ["-6", "--no-padding", "--no-seektable"],
    ["-8", "-S", "10s", "-S", "0.5s", "-T", "ARTIST=x", "-T", "TITLE=żółw", "-P", "1000"], ["-3", "--lax", "-b", "577"],
    ["-8", "-A", "hann", "-A", "tukey(0.25)"], ["-7", "--skip=1000", "--until=-2000"], ["-2", "--replay-gain"],
], ids=lambda a: " ".join(a))
def test_reference_tool_on_this_library_writes_the_reference_tools_files(gpu_engine, tmp_path, args):
    _need(TOOL_REF, TOOL_AMD)
    pcm = synth.music_like(6.0, seed=5)
    pcm[50000:90000] >>= 2
    pcm[50000:90000] <<= 2
    _wav(str(tmp_path / "in.wav"), pcm, 44100)
    ra = _tool(TOOL_REF, ["--silent", "-f"] + args + ["-o", "ref.flac", "in.wav"], str(tmp_path))
    rb = _tool(TOOL_AMD, ["--silent", "-f"] + args + ["-o", "amd.flac", "in.wav"], str(tmp_path))
    assert ra[0] == 0, ra[2]
    assert rb[0] == 0, rb[2]
    assert open(tmp_path / "ref.flac", "rb").read() == open(tmp_path / "amd.flac", "rb").read()


def test_reference_tool_on_this_library_codes_24_bit_multichannel(gpu_engine, tmp_path):
    _need(TOOL_REF, TOOL_AMD)
    rng = np.random.default_rng(3)
    t = np.arange(96000 * 2)
    pcm = np.stack([(3e6 * np.sin(2 * np.pi * (220.0 * (c + 1)) * t / 96000.0)).astype(np.int64) + rng.integers(-3000, 3000, t.size) for c in range(2)], axis=1).astype(np.int32)
    _wav24(str(tmp_path / "in.wav"), pcm, 96000)
    for exe, out in ((TOOL_REF, "ref.flac"), (TOOL_AMD, "amd.flac")):
        rc, _, err = _tool(exe, ["--silent", "-f", "-8", "-o", out, "in.wav"], str(tmp_path))
        assert rc == 0, err
    assert open(tmp_path / "ref.flac", "rb").read() == open(tmp_path / "amd.flac", "rb").read()


def _analysis_input(kind, path):
    rng = np.random.default_rng(17)
    if kind == "music16":
        _wav(path, synth.music_like(5.0, seed=8), 44100)
    elif kind == "noise16":         # verbatim subframes
        _wav(path, rng.integers(-32768, 32768, (60000, 2)).astype(np.int32), 44100)
    elif kind == "wasted-const16":  # wasted bits, a constant channel, a silent stretch
        pcm = synth.music_like(4.0, seed=12)
        pcm[20000:70000] >>= 4
        pcm[20000:70000] <<= 4
        pcm[90000:120000, 0] = -77
        pcm[130000:150000] = 0
        _wav(path, pcm, 44100)
    elif kind == "wide24":          # 24-bit: predictions summed in 64 bits
        t = np.arange(96000)
        pcm = np.stack([(5e6 * np.sin(2 * np.pi * 440.0 * (c + 1) * t / 96000.0)).astype(np.int64) + rng.integers(-90000, 90000, t.size) for c in range(2)], axis=1).astype(np.int32)
        _wav24(path, pcm, 96000)
    return path


@pytest.mark.parametrize("kind,level", [("music16", "-8"), ("music16", "-0"), ("music16", "-3"), ("noise16", "-5"), ("wasted-const16", "-8"), ("wide24", "-8"), ("wide24", "-5")])
def test_reference_tool_on_this_library_decodes_tests_and_analyzes(gpu_engine, tmp_path, kind, level):
    """-d, -t and the analysis listing with every residual of every subframe (src/flac/analyze.c:61-125 reads
    FLAC__Subframe_*::residual and FLAC__Subframe_Verbatim::data in the decoder's write callback)."""
    _need(TOOL_REF, TOOL_AMD)
    _analysis_input(kind, str(tmp_path / "in.wav"))
    assert _tool(TOOL_REF, ["--silent", "-f", level, "-o", "in.flac", "in.wav"], str(tmp_path))[0] == 0
    outs = {}
    for tag, exe in (("ref", TOOL_REF), ("amd", TOOL_AMD)):
        rc, _, err = _tool(exe, ["--silent", "-f", "-d", "-o", tag + ".wav", "in.flac"], str(tmp_path))
        assert rc == 0, err
        rc, _, err = _tool(exe, ["--silent", "-f", "-d", "--force-aiff-format", "--skip=0:00.5", "--until=0:01", "-o", tag + ".aiff", "in.flac"], str(tmp_path))
        assert rc == 0, err
        rc, _, err = _tool(exe, ["--silent", "-t", "in.flac"], str(tmp_path))
        assert rc == 0, err
        rc, _, err = _tool(exe, ["--silent", "-f", "-a", "--residual-text", "-o", tag + ".ana", "in.flac"], str(tmp_path))
        assert rc == 0, err
        outs[tag] = [open(tmp_path / (tag + ext), "rb").read() for ext in (".wav", ".aiff", ".ana")]
    assert outs["ref"][0] == outs["amd"][0] == open(tmp_path / "in.wav", "rb").read()
    assert outs["ref"][1] == outs["amd"][1]
    assert outs["ref"][2] == outs["amd"][2]        # every residual of every subframe
    if kind == "noise16":
        assert b"type=VERBATIM" in outs["amd"][2]
    if level == "-0":
        assert b"type=FIXED" in outs["amd"][2]


def test_decoder_without_the_residual_pass_hands_out_null_residuals(gpu_engine, tmp_path):
    """FLACAMD_DECODER_RESIDUALS=0 (or flacamd_stream_decoder_set_residuals(decoder, 0)): same samples, same listing of
    everything but the residuals."""
    _need(TOOL_REF, TOOL_AMD)
    _analysis_input("music16", str(tmp_path / "in.wav"))
    assert _tool(TOOL_REF, ["--silent", "-f", "-5", "-o", "in.flac", "in.wav"], str(tmp_path))[0] == 0
    env = dict(os.environ, FLACAMD_DECODER_RESIDUALS="0")
    for tag, exe, e in (("ref", TOOL_REF, None), ("amd", TOOL_AMD, env)):
        r = subprocess.run([exe, "--silent", "-f", "-a", "-o", tag + ".ana", "in.flac"], capture_output=True, text=True, timeout=300, cwd=str(tmp_path), env=e)
        assert r.returncode == 0, r.stderr
        r = subprocess.run([exe, "--silent", "-f", "-d", "-o", tag + ".wav", "in.flac"], capture_output=True, text=True, timeout=300, cwd=str(tmp_path), env=e)
        assert r.returncode == 0, r.stderr
    assert open(tmp_path / "ref.ana", "rb").read() == open(tmp_path / "amd.ana", "rb").read()
    assert open(tmp_path / "ref.wav", "rb").read() == open(tmp_path / "amd.wav", "rb").read()


def test_reference_tool_on_this_library_reports_a_damaged_file_like_the_reference(gpu_engine, tmp_path):
    _need(TOOL_REF, TOOL_AMD)
    pcm = synth.music_like(3.0, seed=9)
    _wav(str(tmp_path / "in.wav"), pcm, 44100)
    assert _tool(TOOL_REF, ["--silent", "-f", "-5", "-o", "in.flac", "in.wav"], str(tmp_path))[0] == 0
    data = bytearray(open(tmp_path / "in.flac", "rb").read())
    data[len(data) // 2] ^= 0x10
    open(tmp_path / "bad.flac", "wb").write(bytes(data))
    res = [_tool(exe, ["-t", "bad.flac"], str(tmp_path)) for exe in (TOOL_REF, TOOL_AMD)]
    assert res[0][0] == res[1][0] != 0
    assert res[0][2] == res[1][2]
