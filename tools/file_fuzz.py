"""Development aid: random WAVE / AIFF files through the file front end and through the reference's flac tool
(oracle/_ref/flac, built from the reference's sources): encoded files and decoded files must be identical."""
import os, subprocess, sys, tempfile, zlib
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import oracle_bindings as ob
import test_gpu_files as tf
import parity_fuzz as pf
from flac_amd import files

def main():
    ncase = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    bad = 0; ran = 0
    tmp = tempfile.mkdtemp()
    for seed in range(ncase):
        rng = np.random.default_rng(zlib.crc32(b"ffuzz%d" % seed))
        ch = int(rng.choice([1, 2, 2, 3, 6])); bps = int(rng.choice([8, 16, 16, 24])); level = int(rng.integers(0, 9))
        rate = int(rng.choice([8000, 22050, 44100, 48000, 96000]))
        n = int(rng.integers(1, 60000))
        pcm = pf.signal(rng, int(rng.integers(6)), n, ch, bps)
        aiff = bool(rng.integers(3) == 0) and bps != 8
        ext = ch > 2
        blob = tf.aiff_bytes(pcm, rate, bps) if aiff else tf.wave_bytes(pcm, rate, bps, extensible=ext, mask=(1 << ch) - 1 if ext else 0)
        src = os.path.join(tmp, "in.aiff" if aiff else "in.wav")
        open(src, "wb").write(blob)
        tool, kw = [], {}
        if rng.integers(3) == 0:
            tool += ["-V"]; kw["verify"] = True
        if rng.integers(3) == 0:
            b = int(rng.choice([576, 1152, 2304, 4096, 4608])); tool += ["-b", str(b)]; kw["blocksize"] = b
        if rng.integers(4) == 0:
            tool += ["-S", "3x"]; kw["seekpoints"] = "3x;"
        if rng.integers(4) == 0:
            tool += ["-T", "A=%d" % seed]; kw["tags"] = ["A=%d" % seed]
        want, got = os.path.join(tmp, "ref.flac"), os.path.join(tmp, "gpu.flac")
        try:
            r = subprocess.run([ob.REF_FLAC, "--silent", "-f", "-%d" % level] + tool + ["-o", want, src], capture_output=True, timeout=120)
            if r.returncode != 0:
                continue
            ran += 1
            files.encode_file(src, got, level=level, batch_frames=int(rng.integers(1, 20)), **kw)
            same = open(got, "rb").read() == open(want, "rb").read()
            if same:
                fmt, name = (files.AIFF, "d.aiff") if rng.integers(2) else (files.WAVE, "d.wav")
                w2, g2 = os.path.join(tmp, "ref_" + name), os.path.join(tmp, "gpu_" + name)
                subprocess.run([ob.REF_FLAC, "--silent", "-f", "-d"] + (["--force-aiff-format"] if fmt == files.AIFF else []) + ["-o", w2, want], check=True, timeout=120)
                files.decode_file(want, g2, output_format=fmt)
                same = open(g2, "rb").read() == open(w2, "rb").read()
        except Exception as ex:
            same = False
            print("seed", seed, "exception", repr(ex)[:200])
        if not same:
            bad += 1
            print("seed", seed, "differs: ch", ch, "bps", bps, "rate", rate, "level", level, tool, "aiff", aiff, "n", n, flush=True)
    print("cases", ncase, "ran", ran, "differing", bad)

if __name__ == "__main__":
    main()
