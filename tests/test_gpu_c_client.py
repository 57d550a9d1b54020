"""Compiled C clients of the boundary: the same client sources (examples/c/*.c, written against the libFLAC API)
are built against include/flac_amd_stream.h + libflacamd.so and against the reference's own headers and library;
run on the same input both must write the same files.  The decode client reads every field of FLAC__Frame
(header, subframes[], footer), so this also pins the struct layout a C client sees."""
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


@pytest.mark.parametrize("level,seconds,ch", [(5, 3.0, 2), (8, 1.7, 2), (0, 2.0, 1), (4, 2.5, 2)])
def test_encode_client_writes_the_reference_clients_file(gpu_engine, tmp_path, level, seconds, ch):
    _need(os.path.join(OURS, "encode_file"), os.path.join(REF, "client_encode_file"))
    pcm = synth.music_like(seconds, seed=31 + level)[:, :ch]
    wav = str(tmp_path / "in.wav")
    _wav(wav, pcm, 44100)
    a, b = str(tmp_path / "ours.flac"), str(tmp_path / "ref.flac")
    r1 = subprocess.run([os.path.join(OURS, "encode_file"), wav, a, str(level)], capture_output=True, text=True, timeout=120)
    r2 = subprocess.run([os.path.join(REF, "client_encode_file"), wav, b, str(level)], capture_output=True, text=True, timeout=120)
    assert r1.returncode == 0, r1.stderr
    assert r2.returncode == 0, r2.stderr
    assert r1.stdout.replace("ours.flac", "x") == r2.stdout.replace("ref.flac", "x")
    assert open(a, "rb").read() == open(b, "rb").read()


@pytest.mark.parametrize("level,extra", [(8, []), (5, []), (0, []), (8, ["-b", "1152", "-r", "8"]), (3, ["--lax", "-b", "577"])])
def test_decode_client_reads_frames_like_the_reference_client(gpu_engine, tmp_path, level, extra):
    _need(os.path.join(OURS, "decode_analyze"), os.path.join(REF, "client_decode_analyze"), os.path.join(REF, "flac"))
    pcm = synth.music_like(4.0, seed=70 + level)
    pcm[10000:30000] >>= 3
    pcm[10000:30000] <<= 3           # wasted bits
    pcm[40000:60000, 1] = 1234       # a constant channel
    wav = str(tmp_path / "in.wav")
    _wav(wav, pcm, 44100)
    flac = str(tmp_path / "in.flac")
    subprocess.run([os.path.join(REF, "flac"), "--silent", "-f", "-%d" % level] + extra + ["-o", flac, wav], check=True, timeout=120)
    outs = []
    for exe, tag in ((os.path.join(OURS, "decode_analyze"), "ours"), (os.path.join(REF, "client_decode_analyze"), "ref")):
        w, a = str(tmp_path / (tag + ".wav")), str(tmp_path / (tag + ".ana"))
        r = subprocess.run([exe, flac, w, a], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, (tag, r.stderr)
        outs.append((open(w, "rb").read(), open(a).read()))
    assert outs[0][0] == outs[1][0] == open(wav, "rb").read()
    assert outs[0][1] == outs[1][1]
    assert "type=LPC" in outs[0][1] or level == 0
