"""Seeded synthetic PCM generators (numpy only; no GPU, no reference code).

The families follow the shapes the reference's own stream generator exercises
(src/test_streams/main.c:351-573,962-1077: full-scale deflection patterns, wasted bits,
sines, noise) plus a music-like mix that lands near the compression ratio of real CD audio.
All generators return int32 arrays of shape (samples, channels), values right-justified
at `bps` bits like the buffers FLAC__stream_encoder_process_interleaved takes
(include/FLAC/stream_encoder.h:1730-1760).
"""
import numpy as np


def _clip(x, bps):
    lim = (1 << (bps - 1)) - 1
    return np.clip(np.rint(x), -lim - 1, lim).astype(np.int32)


def _coloured_noise(rng, n, a=0.97):
    """One-pole low-passed Gaussian noise, computed blockwise (exactly the recursion
    y[i] = a*y[i-1] + w[i]) with scipy-free numpy: uses cumulative products per block."""
    w = rng.standard_normal(n)
    # y[i] = sum_k a^(i-k) w[k]; do it in blocks of 4096 to keep powers well scaled
    out = np.empty(n)
    carry = 0.0
    B = 2048
    pw = a ** np.arange(B + 1)
    for s in range(0, n, B):
        blk = w[s:s + B]
        m = len(blk)
        # scaled cumulative sum: y[i] = a^i * (carry*a + sum_{k<=i} w[k]/a^k)
        acc = np.cumsum(blk / pw[:m])
        y = pw[:m] * (acc + carry * a)
        out[s:s + m] = y
        carry = y[-1]
    return out


def music_like(seconds, rate=44100, seed=1234, bps=16, peak=20000.0, events=True):
    """Stereo programme: three partials (one with slow FM) under slow AM plus coloured
    noise; R = 0.6 L + 0.4 independent.  With events=True a few seeded sections are made
    hard-panned, anti-phase, silent, constant, bursty or white so that every subframe type,
    channel assignment and partition order is exercised."""
    rng = np.random.default_rng(seed)
    n = int(round(seconds * rate))
    t = np.arange(n) / rate

    def voice(r):
        am = 0.55 + 0.45 * np.sin(2 * np.pi * 0.11 * t + r.uniform(0, 6))
        fm = 0.05 * np.sin(2 * np.pi * 0.3 * t)
        x = (np.sin(2 * np.pi * 441.0 * t)
             + 0.5 * np.sin(2 * np.pi * 1323.7 * t * (1.0 + 0.0) + fm * 40.0)
             + 0.25 * np.sin(2 * np.pi * 5512.5 * t + r.uniform(0, 6)))
        return am * x / 1.75 + 0.18 * _coloured_noise(r, n) / 4.0

    L = voice(rng)
    R = 0.6 * L + 0.4 * voice(rng)
    x = np.stack([L, R], axis=1) * peak
    if events and n >= 8 * 4096:
        nseg = n // 4096
        kinds = ["pan_l", "pan_r", "anti", "silence", "const", "burst", "white", "wasted", "mono"]
        picks = rng.choice(nseg, size=min(nseg // 6 + 1, 4000), replace=False)
        for j, seg in enumerate(np.sort(picks)):
            k = kinds[j % len(kinds)]
            a, b = seg * 4096, min((seg + 1) * 4096, n)
            if k == "pan_l":
                x[a:b, 1] = 0.0
            elif k == "pan_r":
                x[a:b, 0] = 0.0
            elif k == "anti":
                x[a:b, 1] = -x[a:b, 0] + rng.standard_normal(b - a) * 3.0
            elif k == "silence":
                x[a:b, :] = 0.0
            elif k == "const":
                x[a:b, 0] = 1234.0
                x[a:b, 1] = -77.0
            elif k == "burst":
                m = b - a
                pos = rng.integers(0, m - 64)
                env = np.zeros(m)
                env[pos:pos + 64] = 1.0
                x[a:b, :] += (env * rng.standard_normal(m) * peak * 0.6)[:, None]
            elif k == "white":
                x[a:b, :] = rng.uniform(-peak * 1.6, peak * 1.6, size=(b - a, 2))
            elif k == "wasted":
                x[a:b, :] = np.rint(x[a:b, :] / 16.0) * 16.0
            elif k == "mono":
                x[a:b, 1] = x[a:b, 0]
    return _clip(x, bps)


def multichannel_24(seconds, rate=96000, channels=8, seed=4321, scale=4.0e6):
    """24-bit, per-channel tone pairs (220(c+1) Hz, 3001.3(1+0.01c) Hz) under AM plus
    coloured noise."""
    rng = np.random.default_rng(seed)
    n = int(round(seconds * rate))
    t = np.arange(n) / rate
    cols = []
    for c in range(channels):
        am = 0.6 + 0.4 * np.sin(2 * np.pi * (0.07 + 0.01 * c) * t + c)
        x = am * (0.6 * np.sin(2 * np.pi * 220.0 * (c + 1) * t) + 0.3 * np.sin(2 * np.pi * 3001.3 * (1 + 0.01 * c) * t))
        x = x + 0.05 * _coloured_noise(rng, n) / 4.0 + 0.0005 * rng.standard_normal(n)
        cols.append(x * scale)
    return _clip(np.stack(cols, axis=1), 24)


def sine(n, channels=1, bps=16, f1=441.0, f2=0.0, rate=44100, amp=0.9):
    t = np.arange(n) / rate
    full = (1 << (bps - 1)) - 1
    x = np.sin(2 * np.pi * f1 * t) + (np.sin(2 * np.pi * f2 * t) if f2 else 0.0)
    x = x / (2.0 if f2 else 1.0) * amp * full
    return _clip(np.repeat(x[:, None], channels, axis=1) * (np.array([1.0, -0.7, 0.5, 0.3, 0.9, -0.2, 0.6, 0.1])[:channels]), bps)


def full_scale_pattern(n, channels=1, bps=16, pattern=(1, -1)):
    """Full-scale deflection patterns (src/test_streams/main.c:967-973 family)."""
    hi, lo = (1 << (bps - 1)) - 1, -(1 << (bps - 1))
    pat = np.array([hi if p > 0 else lo for p in pattern], dtype=np.int32)
    x = np.resize(pat, n)
    return np.repeat(x[:, None], channels, axis=1).astype(np.int32)


def white_noise(n, channels=2, bps=16, seed=7):
    rng = np.random.default_rng(seed)
    return rng.integers(-(1 << (bps - 1)), (1 << (bps - 1)), size=(n, channels), dtype=np.int64).astype(np.int32)


def wasted_bits(n, channels=2, bps=16, shift=3, seed=11):
    x = music_like(n / 44100.0, seed=seed, bps=bps, events=False)[:n, :channels]
    return ((x >> shift) << shift).astype(np.int32)


def clip_batch(count, seed=777, min_s=1.0, max_s=30.0):
    """Parameters of a batch of short clips: (seconds, bps, rate, channels, seed)."""
    rng = np.random.default_rng(seed)
    out = []
    for i in range(count):
        out.append((float(rng.uniform(min_s, max_s)), int(rng.choice([16, 24])), int(rng.choice([44100, 48000])),
                    int(rng.choice([1, 2])), int(seed * 1000003 + i)))
    return out
