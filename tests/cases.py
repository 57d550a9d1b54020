"""Shared parity cases: (name, pcm factory, channels, bps, rate, params overrides)."""
import numpy as np

from flac_amd import synth


def _music(seconds, seed, ch=2):
    x = synth.music_like(seconds, seed=seed)
    return x[:, :ch]


def _three():
    m = _music(2.0, 5)
    return np.concatenate([m, m[:, :1] // 2], axis=1)


def parity_cases(small=False):
    s = 0.75 if small else 3.0
    cases = []
    for lvl in range(9):
        cases.append(("st16-level%d" % lvl, lambda s=s, lvl=lvl: _music(s, 100 + lvl), 2, 16, 44100, dict(level=lvl)))
    cases += [
        ("st16-8-p", lambda: _music(s, 21), 2, 16, 44100, dict(level=8, do_qlp_coeff_prec_search=1)),
        ("st16-5-bs1152", lambda: _music(s, 22), 2, 16, 44100, dict(level=5, blocksize=1152)),
        ("st16-8-bs4608", lambda: _music(s, 23), 2, 16, 44100, dict(level=8, blocksize=4608)),
        ("st16-8-bs576", lambda: _music(s / 2, 24), 2, 16, 44100, dict(level=8, blocksize=576)),
        ("st16-8-bs16", lambda: _music(0.1, 25), 2, 16, 44100, dict(level=8, blocksize=16)),
        ("st16-8-bs17", lambda: _music(0.1, 26), 2, 16, 44100, dict(level=8, blocksize=17)),
        ("st16-8-bs33", lambda: _music(0.1, 27), 2, 16, 44100, dict(level=8, blocksize=33)),
        ("st16-8-bs16384", lambda: _music(1.5, 28), 2, 16, 44100, dict(level=8, blocksize=16384)),
        ("st16-8-bs65535", lambda: _music(3.5, 60), 2, 16, 44100, dict(level=8, blocksize=65535)),
        ("st16-8-bs32768-r8", lambda: _music(2.5, 61), 2, 16, 44100, dict(level=8, blocksize=32768, max_residual_partition_order=8)),
        ("st24-6-bs20000", lambda: synth.sine(50000, 2, 24, f2=3000.0), 2, 24, 96000, dict(level=6, blocksize=20000)),
        ("mono16-5-bs40000", lambda: _music(2.0, 62, 1), 1, 16, 44100, dict(level=5, blocksize=40000)),
        ("white24-bs65535", lambda: synth.white_noise(140000, 2, 24, seed=8), 2, 24, 96000, dict(level=8, blocksize=65535)),
        ("white16-bs60000", lambda: synth.white_noise(130000, 2, 16, seed=9), 2, 16, 44100, dict(level=5, blocksize=60000)),
        # full-scale 24-bit noise: a narrow partition sum of the reference wraps, the estimate of a fixed subframe falls far
        # short and the frame comes out at twice the verbatim size (the engine's second pass with large slots)
        ("noise24-wrapped-estimate", lambda: np.random.default_rng(3).integers(-7300000, 7300000, (3000, 2)).astype(np.int32), 2, 24, 48000,
         dict(level=8, blocksize=1000, max_residual_partition_order=4)),
        # tables of the general search kernel beyond LDS unless the candidates go in smaller groups
        ("6ch12-bs16384-p-r8", lambda: (synth.multichannel_24(0.3, channels=6, seed=12)[:20467] >> 12).astype(np.int32), 6, 12, 96000,
         dict(level=8, blocksize=16384, do_qlp_coeff_prec_search=1, max_residual_partition_order=8)),
        ("st16-8-l32", lambda: _music(s / 2, 29), 2, 16, 44100, dict(level=8, max_lpc_order=32)),
        ("st16-8-l32-p", lambda: _music(0.3, 30), 2, 16, 44100, dict(level=8, max_lpc_order=32, do_qlp_coeff_prec_search=1)),
        ("st16-8-r8", lambda: _music(s / 2, 31), 2, 16, 44100, dict(level=8, max_residual_partition_order=8)),
        ("st16-8-r2,4", lambda: _music(s / 2, 32), 2, 16, 44100, dict(level=8, min_residual_partition_order=2, max_residual_partition_order=4)),
        ("st16-8-q15", lambda: _music(s / 2, 33), 2, 16, 44100, dict(level=8, qlp_coeff_precision=15)),
        ("mono16-8", lambda: _music(s, 34, 1), 1, 16, 44100, dict(level=8)),
        ("3ch16-8", _three, 3, 16, 44100, dict(level=8)),
        ("wasted", lambda: synth.wasted_bits(20000), 2, 16, 44100, dict(level=8)),
        ("white", lambda: synth.white_noise(20000), 2, 16, 44100, dict(level=8)),
        ("fsd", lambda: synth.full_scale_pattern(20000, 2), 2, 16, 44100, dict(level=8)),
        ("fsd3", lambda: synth.full_scale_pattern(20000, 2, pattern=(1, 1, -1)), 2, 16, 44100, dict(level=8)),
        ("sine8", lambda: synth.sine(20000, 2, 8), 2, 8, 44100, dict(level=8)),
        ("sine12", lambda: synth.sine(20000, 2, 12, f2=3000.0), 2, 12, 44100, dict(level=8)),
        ("sine20", lambda: synth.sine(20000, 2, 20, f2=3000.0), 2, 20, 48000, dict(level=8)),
        ("sine24-st", lambda: synth.sine(20000, 2, 24, f2=3000.0), 2, 24, 96000, dict(level=8)),
        ("sine24-st-p", lambda: synth.sine(9000, 2, 24, f2=3000.0), 2, 24, 96000, dict(level=8, do_qlp_coeff_prec_search=1)),
        ("silence", lambda: np.zeros((10000, 2), np.int32), 2, 16, 44100, dict(level=8)),
        ("multi-apod", lambda: _music(s / 2, 35), 2, 16, 44100, dict(level=8, apodization=[("tukey", 0.25), ("hann", 0), ("gauss", 0.2), ("welch", 0), ("triangle", 0), ("bartlett", 0), ("bartlett_hann", 0), ("blackman", 0), ("blackman_harris_4term_92db", 0), ("connes", 0), ("flattop", 0), ("hamming", 0), ("kaiser_bessel", 0), ("nuttall", 0), ("rectangle", 0)])),
        ("multi-apod-5", lambda: _music(s / 2, 36), 2, 16, 44100, dict(level=5, apodization=[("tukey", 0.5), ("hann", 0)])),
        ("disable-fixed", lambda: _music(s / 2, 37), 2, 16, 44100, dict(level=8, disable_fixed=1)),
        ("disable-all", lambda: _music(s / 2, 38), 2, 16, 44100, dict(level=8, disable_fixed=1, disable_constant=1, disable_verbatim=1)),
        ("disable-all-l2", lambda: _music(s / 2, 39), 2, 16, 44100, dict(level=2, disable_fixed=1, disable_constant=1, disable_verbatim=1)),
        ("rate12345", lambda: _music(s / 2, 40), 2, 16, 12345, dict(level=5)),
        ("rate11000", lambda: _music(s / 2, 41), 2, 16, 11000, dict(level=5)),
        ("rate44110", lambda: _music(s / 2, 42), 2, 16, 44110, dict(level=5)),
        ("loose-1", lambda: _music(8.0 if not small else 3.0, 43), 2, 16, 44100, dict(level=1)),
        ("loose-4", lambda: _music(8.0 if not small else 3.0, 44), 2, 16, 44100, dict(level=4)),
        ("8ch24-8", lambda: synth.multichannel_24(0.2), 8, 24, 96000, dict(level=8)),
        ("8ch24-8-e-p", lambda: synth.multichannel_24(0.1), 8, 24, 96000, dict(level=8, do_qlp_coeff_prec_search=1)),
    ]
    for n in (1, 2, 3, 4, 5, 12, 13, 4095, 4096, 4097, 8192, 8193):
        cases.append(("len%d" % n, lambda n=n: _music(0.25, 50)[:n], 2, 16, 44100, dict(level=8)))
    return cases
