"""Development aid: random batches of complete streams through flacamd_encode_streams_device against the CPU oracle
(tests/oracle_bindings.py) coding each stream alone."""
import os, sys, zlib
import numpy as np
import torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import oracle_bindings as ob
import parity_fuzz as pf
from flac_amd import engine

def main():
    ncase = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    bad = 0; ran = 0
    for seed in range(ncase):
        rng = np.random.default_rng(zlib.crc32(b"bfuzz%d" % seed))
        ch = int(rng.choice([1, 2, 2, 2, 3])); bps = int(rng.choice([8, 16, 16, 16, 24])); level = int(rng.integers(0, 9))
        bs = int(rng.choice([0, 0, 192, 1152, 4096, 1000, 2304]))
        kw = dict(level=level)
        if bs: kw["blocksize"] = bs
        if rng.integers(5) == 0: kw["max_residual_partition_order"] = int(rng.integers(0, 9))
        try:
            fp, _ = ob.make_params(ch, bps, 44100, **kw)
            params = engine.make_params(ch, bps, 44100, **kw)
            p2 = engine.Params.from_buffer_copy(params)
            if engine.lib().flacamd_params_resolve(p2) != 0:
                continue
            eff = int(p2.blocksize)
            lens = [int(rng.integers(0, 4)) * eff + int(rng.integers(1, eff + 1)) for _ in range(int(rng.integers(1, 12)))]
            streams = [pf.signal(rng, int(rng.integers(6)), n, ch, bps) for n in lens]
            want = []
            for pcm in streams:
                want += ob.oracle_encode(fp, pcm)["frames"]
            dt = np.int16 if (bps <= 16 and rng.integers(2)) else np.int32
            nfr = [-(-n // eff) for n in lens]
            starts = np.cumsum([0] + [f * eff for f in nfr])[:-1]
            host = rng.integers(-3, 3, size=(sum(nfr) * eff, ch)).astype(dt)
            for st, pcm in zip(starts, streams):
                host[st:st + len(pcm)] = pcm.astype(dt)
            d_pcm = torch.from_numpy(host).cuda()
            F = sum(nfr)
            enc = engine.Encoder(params, max_frames=F)
            cap = enc.max_output_bytes(F) * 4
            d_out = torch.empty(cap, dtype=torch.uint8, device="cuda"); d_fb = torch.empty(F, dtype=torch.int32, device="cuda")
            total = enc.encode_streams_device(d_pcm.data_ptr(), engine.PCM_S16 if dt == np.int16 else engine.PCM_S32, starts, lens, d_out.data_ptr(), cap, d_fb.data_ptr())
            enc.close()
            sizes = d_fb.cpu().numpy(); data = d_out[:total].cpu().numpy().tobytes()
            got, off = [], 0
            for s in sizes:
                got.append(data[off:off + int(s)]); off += int(s)
            same = got == want
            ran += 1
        except Exception as ex:
            same = False
            print("seed", seed, "exception", repr(ex)[:200])
        if not same:
            bad += 1
            print("seed", seed, "differs: ch", ch, "bps", bps, kw, "lens", lens, flush=True)
    print("cases", ncase, "ran", ran, "differing", bad)

if __name__ == "__main__":
    main()
