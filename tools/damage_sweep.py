"""Sweep of randomly damaged streams through this library's stream decoder and the reference decoder
(oracle/_ref/libflacref.so): prints the cases where what the client observes differs."""
import os, sys, zlib
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import test_gpu_damaged_streams as t

def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    bad_cases = 0
    for seed in range(n):
        kind = ["st16", "st16-8", "mono24", "wasted"][seed % 4]
        data, audio, sizes = t._stream(kind)
        rng = np.random.default_rng(zlib.crc32(b"sweep%d" % seed))
        bad = bytearray(data)
        mode = seed % 5
        if mode == 0:
            for _ in range(int(rng.integers(1, 8))):
                bad[int(rng.integers(audio, len(bad)))] ^= int(rng.integers(1, 256))
        elif mode == 1:
            a = int(rng.integers(audio, len(bad) - 600)); ln = int(rng.integers(1, 500))
            bad[a:a + ln] = bytes(rng.integers(0, 256, ln, dtype=np.uint8))
        elif mode == 2:
            a = int(rng.integers(audio, len(bad) - 600)); ln = int(rng.integers(1, 500))
            del bad[a:a + ln]
        elif mode == 3:
            a = int(rng.integers(audio, len(bad))); ln = int(rng.integers(1, 300))
            bad[a:a] = bytes(rng.integers(0, 256, ln, dtype=np.uint8))
        else:
            bad = bad[:int(rng.integers(audio + 1, len(bad)))]
            if rng.integers(0, 2):
                bad[int(rng.integers(audio, len(bad)))] ^= 1 << int(rng.integers(0, 8))
        batch = [None, 1, 2, 7][(seed // 5) % 4]
        ref = t._run(t._ref_decoder_lib(), bad)
        got = t._run(None, bad, batch=batch, read_chunk=[None, 1000, 77][seed % 3])
        same = got["events"] == ref["events"] and got["ok"] == ref["ok"] and got["state"] == ref["state"] and got["finish"] == ref["finish"] \
            and got["pcm"].shape == ref["pcm"].shape and np.array_equal(got["pcm"], ref["pcm"])
        if not same:
            bad_cases += 1
            k = next((i for i, (a, b) in enumerate(zip(got["events"], ref["events"])) if a != b), min(len(got["events"]), len(ref["events"])))
            print("seed", seed, kind, "mode", mode, "batch", batch, "first difference at event", k, "got", got["events"][k:k + 3], "ref", ref["events"][k:k + 3],
                  "ok", got["ok"], ref["ok"], "state", got["state"], ref["state"], "finish", got["finish"], ref["finish"], flush=True)
    print("cases", n, "differing", bad_cases)

if __name__ == "__main__":
    main()
