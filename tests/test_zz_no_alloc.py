"""The allocation-free MD5 entry point against the allocating one it is a variant of
(flacamd_md5_device), through the C ABI: two calls in a row on the same GPU, the second with a
smaller batch, so that what the first call left in the fixed staging array cannot leak into
the second; both also checked against hashlib."""
import ctypes as C
import hashlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _static(L, d_ptr, fmt, ch, bps, offs, device=0):
    off = np.ascontiguousarray(offs, dtype=np.uint64)
    n = len(off) - 1
    out = np.full((n, 16), 0xA5, dtype=np.uint8)
    rc = L.flacamd_md5_device_static(device, d_ptr, fmt, ch, bps, off.ctypes.data, n, out.ctypes.data, None)
    assert rc == 0, (rc, L.flacamd_last_error())
    return [bytes(r) for r in out]


def _batch(rng, nstreams, ch, bps, dtype):
    lens = rng.integers(1, 3000, size=nstreams)
    lens[0] = 1
    lens[-1] = 64 * 7          # a stream that is whole blocks only
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    lo, hi = -(1 << (bps - 1)), (1 << (bps - 1)) - 1
    pcm = rng.integers(lo, hi + 1, size=(int(offs[-1]), ch)).astype(dtype)
    return pcm, offs


def _hashlib(pcm, offs, bps):
    nb = (bps + 7) // 8
    want = []
    for a, b in zip(offs[:-1], offs[1:]):
        raw = pcm[int(a):int(b)].astype("<i4").reshape(-1).view(np.uint8).reshape(-1, 4)[:, :nb]
        want.append(hashlib.md5(raw.tobytes()).digest())
    return want


def test_md5_static_matches_allocating_entry_point(gpu_engine):
    import torch
    L = gpu_engine.lib()
    L.flacamd_md5_device_static.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
    L.flacamd_md5_device_static_max_streams.restype = C.c_uint32
    assert L.flacamd_md5_device_static_max_streams() == 65536
    rng = np.random.default_rng(20260)
    # first call: many streams (fills a good part of the staging array); second: fewer, another format
    for nstreams, ch, bps, dtype in ((5000, 2, 16, np.int16), (37, 2, 24, np.int32), (300, 1, 8, np.int32)):
        pcm, offs = _batch(rng, nstreams, ch, bps, dtype)
        d = torch.from_numpy(pcm).cuda()
        fmt = gpu_engine.PCM_S16 if dtype == np.int16 else gpu_engine.PCM_S32
        got = _static(L, d.data_ptr(), fmt, ch, bps, offs)
        ref = gpu_engine.md5_device(d.data_ptr(), fmt, ch, bps, offs)
        assert got == ref
        assert got == _hashlib(pcm, offs, bps)


def test_md5_static_rejects_batches_beyond_its_array(gpu_engine):
    import torch
    L = gpu_engine.lib()
    L.flacamd_md5_device_static.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
    n = 65537
    offs = np.arange(n + 1, dtype=np.uint64)
    d = torch.zeros((n, 1), dtype=torch.int16, device="cuda")
    out = np.zeros((n, 16), np.uint8)
    assert L.flacamd_md5_device_static(0, d.data_ptr(), gpu_engine.PCM_S16, 1, 16, offs.ctypes.data, n, out.ctypes.data, None) == -6
    # the largest batch it takes
    n = 65536
    got = _static(L, d.data_ptr(), gpu_engine.PCM_S16, 1, 16, offs[:n + 1])
    assert got[0] == hashlib.md5(b"\0\0").digest() and got[-1] == got[0]
