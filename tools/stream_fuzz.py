"""Development aid: random configurations through the libFLAC-style encoder object (random chunk sizes, batch sizes,
verify on or off) against the live reference library, and the result back through the decoder object."""
import os, sys, zlib
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import oracle_bindings as ob
import test_gpu_stream as ts
import parity_fuzz as pf
from flac_amd import stream

def main():
    ncase = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    bad = 0
    for seed in range(first, first + ncase):
        rng = np.random.default_rng(zlib.crc32(b"sfuzz%d" % seed))
        ch = int(rng.choice([1, 2, 2, 2, 3, 8])); bps = int(rng.choice([8, 16, 16, 16, 20, 24])); level = int(rng.integers(0, 9))
        bs = int(rng.choice([0, 0, 192, 1152, 4096, 4608, 1000, 16384, 20000]))
        kw = dict(level=level)
        if bs: kw["blocksize"] = bs
        if rng.integers(5) == 0: kw["do_qlp_coeff_prec_search"] = 1
        if rng.integers(5) == 0: kw["max_residual_partition_order"] = int(rng.integers(0, 9))
        eff = bs or (4096 if level >= 3 else 1152)
        n = min(int(rng.integers(0, 5)) * eff + int(rng.integers(1, eff + 1)), 60000)
        pcm = pf.signal(rng, int(rng.integers(6)), n, ch, bps)
        rate = int(rng.choice([8000, 44100, 48000, 96000]))
        try:
            _, rp = ob.make_params(ch, bps, rate, **kw)
            try:
                ref = ob.ref_encode(rp, pcm, chunk=int(rng.integers(1, 5000)))
            except RuntimeError:
                continue
            e = stream.MemoryEncoder(seekable=False)
            ts._configure(e, ch, bps, rate, kw)
            assert e.set_verify(int(rng.integers(2))) and e.set_batch_frames(int(rng.integers(1, 9)))
            assert e.init() == stream.INIT_OK
            ts._feed(e, pcm, int(rng.integers(1, 9000)))
            ok = e.finish()
            same = ok and bytes(e.buf) == ref["stream"]
            e.close()
            if same:
                d = stream.StreamDecoder(ref["stream"], seekable=False, read_chunk=int(rng.integers(1, 20000)))
                assert d.set_batch_frames(int(rng.integers(1, 9))) and d.init() == 0
                same = d.process_until_end_of_stream() and d.errors == [] and np.array_equal(d.pcm(), pcm)
                d.close()
        except Exception as ex:
            same = False
            print("seed", seed, "exception", repr(ex)[:200])
        if not same:
            bad += 1
            print("seed", seed, "differs: ch", ch, "bps", bps, "rate", rate, kw, "n", n, flush=True)
    print("cases", ncase, "differing", bad)

if __name__ == "__main__":
    main()
