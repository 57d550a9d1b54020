"""CPU suite for the product library: it loads, exports every symbol the header declares,
and its host-side helpers (parameter resolution, window tables) agree with the oracle.
No compute entry point is exercised here (there is no GPU)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import oracle_bindings as ob
from flac_amd import engine

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(flacamd_[a-z0-9_]+|FLAC__[A-Za-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = C.CDLL(engine.LIB_PATH)
    names = _declared_symbols("flac_amd_engine.h")
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), n


def test_presets_and_resolution_match_oracle():
    olib = ob.oracle_lib()
    for lvl in range(9):
        for ch, bps, bs in ((2, 16, 0), (1, 16, 1152), (2, 24, 0), (8, 24, 4096), (2, 8, 192), (2, 16, 16384)):
            p = engine.make_params(ch, bps, 44100, level=lvl, blocksize=bs)
            assert engine.lib().flacamd_params_resolve(C.byref(p)) == 0
            fp, _ = ob.make_params(ch, bps, 44100, level=lvl, blocksize=bs)
            assert olib.fo_params_resolve(C.byref(fp)) == 0
            for name, _t in engine.Params._fields_[:16]:
                assert getattr(p, name) == getattr(fp, name), (lvl, ch, bps, bs, name)


def test_resolution_rejects_what_the_reference_rejects():
    L = engine.lib()
    for kw in (dict(channels=0), dict(channels=9), dict(bits_per_sample=3), dict(bits_per_sample=25), dict(sample_rate=0),
               dict(sample_rate=655351), dict(blocksize=15), dict(max_lpc_order=33), dict(qlp_coeff_precision=4), dict(qlp_coeff_precision=16)):
        base = dict(channels=2, bits_per_sample=16, sample_rate=44100)
        base.update(kw)
        p = engine.make_params(base["channels"], base["bits_per_sample"], base["sample_rate"], level=8,
                               **{k: v for k, v in kw.items() if k not in base or k in ("blocksize", "max_lpc_order", "qlp_coeff_precision")})
        assert L.flacamd_params_resolve(C.byref(p)) == -1, kw
    # legal for the reference, outside the engine's limits: reported as such, never silently changed
    p = engine.make_params(2, 16, 44100, level=8, blocksize=32768)
    assert L.flacamd_params_resolve(C.byref(p)) == -2
    p = engine.make_params(2, 16, 44100, level=8, max_residual_partition_order=9)
    assert L.flacamd_params_resolve(C.byref(p)) == -2


def test_host_window_tables_match_oracle():
    olib = ob.oracle_lib()
    for L in (16, 33, 4096, 4097):
        for i, name in enumerate(engine.WINDOWS):
            for prm in ((0.0, 0.3, 0.5, 1.0) if name == "tukey" else (0.1, 0.5) if name == "gauss" else (0.0,)):
                a = engine.window(name, prm, L)
                b = np.zeros(L, np.float32)
                olib.fo_window(i, prm, L, b.ctypes.data)
                assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (name, prm, L)


def test_compute_entry_points_fail_loudly_without_gpu():
    if engine.device_count() > 0:
        pytest.skip("GPU present")
    p = engine.make_params(2, 16, 44100, level=8)
    with pytest.raises(engine.EngineError) as ei:
        engine.Encoder(p, max_frames=4)
    assert ei.value.code == -3
