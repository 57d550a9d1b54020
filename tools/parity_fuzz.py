"""Development aid: random encoder configurations and signals through the device engine and the CPU oracle
(tests/oracle_bindings.py); prints every case whose frames differ."""
import os, sys, zlib
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import oracle_bindings as ob
import test_gpu_parity as tp
from flac_amd import engine, synth

def signal(rng, kind, n, ch, bps):
    peak = (1 << (bps - 1)) - 1
    t = np.arange(n)
    cols = []
    for c in range(ch):
        if kind == 0:
            x = synth.music_like(n / 44100.0 + 0.05, seed=int(rng.integers(1 << 30)))[:n, c % 2].astype(np.float64) / 32768.0 * peak * rng.uniform(0.1, 1.0)
        elif kind == 1:
            x = rng.integers(-peak, peak, n).astype(np.float64) * rng.uniform(0.001, 1.0)
        elif kind == 2:
            x = peak * rng.uniform(0.2, 1.0) * np.sin(t * rng.uniform(0.001, 1.5) + c) + rng.normal(0, rng.uniform(0, 30), n)
        elif kind == 3:
            x = (rng.integers(-peak >> 4, peak >> 4, n) << int(rng.integers(1, 4))).astype(np.float64)
        elif kind == 4:
            x = np.where((t // int(rng.integers(1, 400))) % 2 == 0, peak, -peak - 1).astype(np.float64)
        else:
            x = np.cumsum(rng.normal(0, peak / 200.0, n)) % peak - peak / 2
        cols.append(np.clip(np.round(x), -peak - 1, peak).astype(np.int32))
    if ch == 2 and rng.integers(3) == 0:
        cols[1] = np.clip(cols[0] + rng.integers(-3, 4, n), -peak - 1, peak).astype(np.int32)   # nearly identical channels: side signal near zero
    return np.stack(cols, axis=1)

def case(seed):
    """The seeded case: (channels, bits per sample, rate, encoder settings, pcm, effective blocksize)."""
    rng = np.random.default_rng(zlib.crc32(b"fuzz%d" % seed))
    ch = int(rng.choice([1, 2, 2, 2, 3, 6, 8]))
    bps = int(rng.choice([8, 12, 16, 16, 16, 20, 24]))
    level = int(rng.integers(0, 9))
    bs = int(rng.choice([0, 0, 16, 192, 576, 1000, 1152, 2304, 4096, 4096, 4608, 5000, 8192, 16384, 20000]))
    kw = dict(level=level)
    if bs:
        kw["blocksize"] = bs
    if rng.integers(4) == 0:
        kw["do_qlp_coeff_prec_search"] = 1
    if rng.integers(4) == 0:
        kw["max_lpc_order"] = int(rng.integers(0, 33))
    if rng.integers(4) == 0:
        kw["max_residual_partition_order"] = int(rng.integers(0, 9))
    if rng.integers(6) == 0:
        kw["qlp_coeff_precision"] = int(rng.integers(5, 16))
    if rng.integers(8) == 0:
        kw["do_exhaustive_model_search"] = 1
    if rng.integers(8) == 0:
        kw["apodization"] = [("tukey", 0.25), ("hann", 0), ("blackman", 0)][:int(rng.integers(1, 4))]
    if rng.integers(10) == 0:
        kw["disable_fixed"] = 1
    if rng.integers(10) == 0:
        kw["disable_verbatim"] = 1
    if rng.integers(10) == 0:
        kw["disable_constant"] = 1
    if rng.integers(8) == 0:
        kw["min_residual_partition_order"] = int(rng.integers(0, 5))
    eff = bs or (4096 if kw.get("max_lpc_order", 1 if level >= 3 else 0) else 1152)
    n = int(rng.integers(1, 4)) * eff + int(rng.integers(0, eff))
    n = min(n, 70000)
    pcm = signal(rng, int(rng.integers(6)), n, ch, bps)
    rate = int(rng.choice([8000, 44100, 48000, 96000]))
    return ch, bps, rate, kw, pcm, eff


def run_seed(seed):
    """None when the reference refuses the configuration, else (identical?, description): frames from the device engine
    against the oracle's, the oracle's frames back through the device decoder, and the 16-bit input layout."""
    ch, bps, rate, kw, pcm, eff = case(seed)
    n = len(pcm)
    what = "seed %d: ch %d bps %d rate %d %r n %d" % (seed, ch, bps, rate, kw, n)
    fp, _ = ob.make_params(ch, bps, rate, **kw)
    try:
        want = ob.oracle_encode(fp, pcm)["frames"]
    except RuntimeError:
        return None   # a configuration the reference refuses (e.g. blocksize <= max_lpc_order)
    enc, (frames, _) = tp._encode(engine, ch, bps, rate, kw, pcm)
    enc.close()
    if frames != want:
        return False, what + " (frames differ)"
    blob = b"".join(want)
    dec = engine.Decoder(ch, bps, max(eff, 16), max_frames=len(want) + 64, max_bytes=len(blob) + 64)
    back, info = dec.decode_host(blob, n + eff)
    dec.close()
    if info["frames"] != len(want) or not np.array_equal(back, pcm):
        return False, what + " (decoder differs)"
    if bps <= 16:
        enc, (frames16, _) = tp._encode(engine, ch, bps, rate, kw, pcm.astype(np.int16))
        enc.close()
        if frames16 != want:
            return False, what + " (16-bit input layout differs)"
    return True, what


def main():
    ncase = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    bad = 0
    for seed in range(first, first + ncase):
        try:
            r = run_seed(seed)
        except Exception as ex:
            r = (False, "seed %d exception %s" % (seed, repr(ex)[:200]))
        if seed % 500 == 499:
            print("... seed", seed, "differing so far", bad, flush=True)
        if r is not None and not r[0]:
            bad += 1
            print(r[1], flush=True)
    print("cases", ncase, "differing", bad)

if __name__ == "__main__":
    main()
