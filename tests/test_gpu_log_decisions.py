"""The one operation on the encoder path that IEEE 754 does not pin is log(): fixed.c:265 and lpc.c:1339 call the C
library's, the model stage calls the device's.  Every decision downstream of a logarithm is a threshold on its result
(the Rice parameter suggested to a candidate rounds it to an integer, the "not worth trying" skip compares it with the
sample size, the order guess compares sums of such terms), so the two libraries only have to agree in the neighbourhood
of those thresholds.  Here that neighbourhood is swept through the device's own expressions (flacamd_debug_model_logs)
and compared with the host's libm, which is what the reference runs on."""
import ctypes as C
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
LN2 = 0.69314718055994530942


def _device_logs(engine, e, length, err, scale):
    L = engine.lib()
    L.flacamd_debug_model_logs.argtypes = [C.c_int, C.c_uint32] + [C.c_void_p] * 6
    n = len(e)
    e = np.ascontiguousarray(e, np.uint64)
    length, err, scale = (np.ascontiguousarray(a, np.float64) for a in (length, err, scale))
    fb = np.zeros(n, np.float32)
    eb = np.zeros(n, np.float64)
    rc = L.flacamd_debug_model_logs(0, n, e.ctypes.data, length.ctypes.data, err.ctypes.data, scale.ctypes.data, fb.ctypes.data, eb.ctypes.data)
    assert rc == 0
    return fb, eb


def _host_fixed(e, length):
    out = np.zeros(len(e), np.float32)
    for i, (a, b) in enumerate(zip(e.tolist(), length.tolist())):
        out[i] = np.float32(math.log(LN2 * float(a) / b) / LN2) if a > 0 else np.float32(0.0)
    return out


def _host_lpc(err, scale):
    out = np.zeros(len(err), np.float64)
    for i, (a, s) in enumerate(zip(err.tolist(), scale.tolist())):
        if a > 0.0:
            b = 0.5 * math.log(s * a) / LN2
            out[i] = b if b >= 0.0 else 0.0
        else:
            out[i] = 1e32 if a < 0.0 else 0.0
    return out


def _fixed_decisions(fb, bps):
    """stream_encoder.c:3254-3263: skip when the estimate reaches the sample size, else the parameter (float -> double, + 0.5, truncate) + 1"""
    k = np.where(fb > 0, (fb.astype(np.float64) + 0.5).astype(np.int64), 0) + 1
    return (fb >= np.float32(bps)), k


def test_fixed_predictor_estimates_around_every_threshold(gpu_engine):
    """The inputs of fixed.c:265 are integers (|residual| sum, block length - 4), so the values next to a threshold can be
    listed: for block lengths across the format's range and every threshold j + 0.5 (parameter rounding) and j (the
    skip, j = sample size), the integers e on either side of the point where the estimate crosses it."""
    lens, es = [], []
    block_lengths = sorted(set([16, 17, 33, 192, 576, 577, 1152, 2304, 4095, 4096, 4097, 4608, 8192, 16384, 32768, 65535] + list(range(20, 65535, 997))))
    for n in block_lengths:
        length = n - 4
        for j2 in range(0, 66):                       # thresholds 0, 0.5, 1, ... 32.5 bits
            centre = length * (2.0 ** (j2 / 2.0)) / LN2
            if centre >= 2.0 ** 63:
                continue
            c = int(centre)
            for d in range(-3, 4):
                if c + d > 0:
                    lens.append(float(length))
                    es.append(c + d)
    e = np.array(es, dtype=np.uint64)
    length = np.array(lens, dtype=np.float64)
    fb, _ = _device_logs(gpu_engine, e, length, np.ones(len(e)), np.ones(len(e)))
    want = _host_fixed(e, length)
    assert len(e) > 30000
    # the stored float itself is equal: every comparison and rounding made of it then is
    assert np.array_equal(fb.view(np.uint32), want.view(np.uint32)), "device and host log differ next to a threshold: %d of %d" % (int((fb.view(np.uint32) != want.view(np.uint32)).sum()), len(e))
    for bps in (8, 16, 17, 24, 25):
        s1, k1 = _fixed_decisions(fb, bps)
        s2, k2 = _fixed_decisions(want, bps)
        assert np.array_equal(s1, s2) and np.array_equal(k1, k2)


def test_lpc_estimates_around_every_threshold(gpu_engine):
    """lpc.c:1339 takes a double prediction error: thresholds can be approached to the last bit, so each one (j + 0.5 for
    the parameter, j for the skip) is bracketed by the doubles within a few units in the last place on either side, for
    residual counts across the range; the decisions made of the result must not differ."""
    errs, scales = [], []
    for count in (12, 100, 1140, 4084, 4095, 16372, 65503):
        scale = 0.5 * LN2 * LN2 / count
        for j2 in range(0, 66):
            centre = (4.0 ** (j2 / 2.0)) / scale            # 0.5 * log2(scale * err) = j2 / 2
            x = centre
            for _ in range(40):
                x = math.nextafter(x, 0.0)
            for _ in range(81):
                errs.append(x)
                scales.append(scale)
                x = math.nextafter(x, math.inf)
    err = np.array(errs)
    scale = np.array(scales)
    _, eb = _device_logs(gpu_engine, np.ones(len(err), np.uint64), np.ones(len(err)), err, scale)
    want = _host_lpc(err, scale)
    assert len(err) > 30000
    # stream_encoder.c:3331-3339: skip when eb >= sample size; parameter = (eb > 0 ? (unsigned)(eb + 0.5) : 0) + 1
    for bps in (8, 16, 17, 24, 25):
        assert np.array_equal(eb >= bps, want >= bps)
    k1 = np.where(eb > 0, (eb + 0.5).astype(np.int64), 0)
    k2 = np.where(want > 0, (want + 0.5).astype(np.int64), 0)
    assert np.array_equal(k1, k2), "suggested parameters differ for %d of %d inputs" % (int((k1 != k2).sum()), len(err))
    # how far apart the two libraries are at all, in units in the last place (for the record: this is what the thresholds have to absorb)
    ulp = np.abs(eb.view(np.int64) - want.view(np.int64))
    assert ulp.max() <= 2
