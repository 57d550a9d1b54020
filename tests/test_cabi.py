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
    text = re.sub(r"^\s*#\s*define.*$", "", text, flags=re.M)   # object-like macros are not symbols
    return sorted(set(re.findall(r"\b(flacamd_[a-z0-9_]+|FLAC__[A-Za-z0-9_]+)\s*\((?!\*)", text)))


def test_library_exports_every_declared_symbol():
    lib = C.CDLL(engine.LIB_PATH)
    names = _declared_symbols("flac_amd_engine.h")
    assert len(names) >= 20
    stream_names = _declared_symbols("flac_amd_stream.h")
    assert len(stream_names) >= 50 and "FLAC__stream_encoder_process_interleaved" in stream_names
    assert "FLAC__stream_decoder_process_until_end_of_stream" in stream_names and "flacamd_decoder_frame_table" in names
    for n in names + stream_names + ["FLAC__StreamEncoderStateString", "FLAC__StreamEncoderInitStatusString", "FLAC__StreamDecoderStateString",
                                     "FLAC__StreamDecoderInitStatusString", "FLAC__StreamDecoderErrorStatusString"]:
        assert hasattr(lib, n), n


def test_presets_and_resolution_match_oracle():
    olib = ob.oracle_lib()
    for lvl in range(9):
        for ch, bps, bs in ((2, 16, 0), (1, 16, 1152), (2, 24, 0), (8, 24, 4096), (2, 8, 192), (2, 16, 16384), (2, 16, 32768), (2, 24, 65535)):
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


# ---- the libFLAC-style object: everything up to the point where a device is needed ----------

def test_stream_encoder_defaults_setters_and_getters():
    from flac_amd import stream
    e = stream.StreamEncoder()
    # set_defaults_ (stream_encoder.c:2107-2160) followed by compression level 5
    assert e.get_state() == stream.UNINITIALIZED and e.get_resolved_state_string() == "FLAC__STREAM_ENCODER_UNINITIALIZED"
    assert (e.get_channels(), e.get_bits_per_sample(), e.get_sample_rate(), e.get_blocksize()) == (2, 16, 44100, 0)
    assert (e.get_streamable_subset(), e.get_verify(), e.get_do_mid_side_stereo(), e.get_loose_mid_side_stereo()) == (1, 0, 1, 0)
    assert (e.get_max_lpc_order(), e.get_qlp_coeff_precision(), e.get_max_residual_partition_order()) == (8, 0, 5)
    fp, _ = ob.make_params(2, 16, 44100, level=0)
    for lvl in range(11):
        assert e.set_compression_level(lvl)
        fp, _ = ob.make_params(2, 16, 44100, level=min(lvl, 8))
        got = (e.get_do_mid_side_stereo(), e.get_loose_mid_side_stereo(), e.get_max_lpc_order(), e.get_qlp_coeff_precision(),
               e.get_do_qlp_coeff_prec_search(), e.get_do_exhaustive_model_search(), e.get_min_residual_partition_order(),
               e.get_max_residual_partition_order())
        want = (fp.do_mid_side, fp.loose_mid_side, fp.max_lpc_order, fp.qlp_coeff_precision, fp.do_qlp_coeff_prec_search,
                fp.do_exhaustive_model_search, fp.min_residual_partition_order, fp.max_residual_partition_order)
        assert got == want, lvl
    assert e.set_channels(3) and e.get_channels() == 3
    assert e.set_blocksize(576) and e.get_blocksize() == 576
    assert e.set_total_samples_estimate(1 << 34) and e.get_total_samples_estimate() == 1 << 34
    # accepted and inert, as in the reference (stream_encoder.c:1635-1649, :1684-1698)
    assert e.set_do_escape_coding(1) and e.get_do_escape_coding() == 0
    assert e.set_rice_parameter_search_dist(3) and e.get_rice_parameter_search_dist() == 0
    assert e.set_verify(1) and e.get_verify() == 1 and e.set_verify(0) and e.get_verify() == 0
    assert e.get_verify_decoder_state() == stream.DEC_UNINITIALIZED   # verify off (:1788-1797)
    assert e.get_verify_decoder_error_stats() == dict(absolute_sample=0, frame_number=0, channel=0, sample=0, expected=0, got=0)
    assert e.set_metadata([stream.padding_block(16)]) and e.set_metadata([])
    assert e.set_apodization("hann;tukey(0.25);nonsense;gauss(0.2)")
    assert e.finish()  # finish on an uninitialised encoder is a successful no-op (:1314-1315)
    e.close()


@pytest.mark.parametrize("setup,status", [
    (dict(channels=0), 4), (dict(channels=9), 4), (dict(bits_per_sample=3), 5), (dict(bits_per_sample=25), 5),
    (dict(sample_rate=0), 6), (dict(sample_rate=655351), 6), (dict(blocksize=15), 7), (dict(blocksize=65536), 7),
    (dict(max_lpc_order=33), 8), (dict(blocksize=16, max_lpc_order=17, streamable_subset=0), 10),
    (dict(qlp_coeff_precision=4), 9), (dict(qlp_coeff_precision=16), 9),
    (dict(blocksize=4609), 11), (dict(sample_rate=96000, blocksize=16385), 11), (dict(sample_rate=65537), 11),
    (dict(bits_per_sample=17), 11), (dict(max_residual_partition_order=9), 11), (dict(max_lpc_order=13), 11),
])
def test_stream_encoder_init_status_codes(setup, status):
    """The order and the codes of init_stream_internal_'s checks (stream_encoder.c:656-766)."""
    from flac_amd import stream
    e = stream.MemoryEncoder()
    for k, v in setup.items():
        assert getattr(e, "set_" + k)(v)
    assert e.init() == status
    assert e.get_state() == stream.UNINITIALIZED and e.writes == []
    e.close()


def test_stream_encoder_requires_a_write_callback_and_a_device():
    from flac_amd import stream
    L = engine.lib()
    e = stream.StreamEncoder()
    assert L.FLAC__stream_encoder_init_stream(e._h, stream.WRITE_CB(), stream.SEEK_CB(), stream.TELL_CB(), stream.META_CB(), None) == 3
    cb = stream.WRITE_CB(lambda *a: 0)
    assert L.FLAC__stream_encoder_init_stream(e._h, cb, stream.SEEK_CB(lambda *a: 0), stream.TELL_CB(), stream.META_CB(), None) == 3
    if engine.device_count() == 0:
        # no silent CPU path: without a HIP device init fails and says so
        m = stream.MemoryEncoder()
        assert m.init() == stream.INIT_ENCODER_ERROR and m.get_state() == stream.MEMORY_ALLOCATION_ERROR
        assert not m.process_interleaved(np.zeros((10, 2), np.int32))
        m.close()
    e.close()


def test_stream_decoder_init_checks_and_no_cpu_path():
    """init_stream_internal_'s callback checks (stream_decoder.c:364-395); no device, no decoder."""
    from flac_amd import stream
    L = engine.lib()
    d = stream.StreamDecoder(b"", seekable=False)
    assert d.get_state() == stream.DEC_UNINITIALIZED and d.get_resolved_state_string() == "FLAC__STREAM_DECODER_UNINITIALIZED"
    none = (stream.D_READ_CB(), stream.D_SEEK_CB(), stream.D_TELL_CB(), stream.D_LENGTH_CB(), stream.D_EOF_CB(), stream.D_WRITE_CB(), stream.D_META_CB(), stream.D_ERROR_CB())
    assert L.FLAC__stream_decoder_init_stream(d._h, *none, None) == stream.DEC_INIT_INVALID_CALLBACKS
    rd, wr, er = stream.D_READ_CB(lambda *a: 1), stream.D_WRITE_CB(lambda *a: 0), stream.D_ERROR_CB(lambda *a: None)
    # a seek callback without tell / length / eof is refused
    assert L.FLAC__stream_decoder_init_stream(d._h, rd, stream.D_SEEK_CB(lambda *a: 0), stream.D_TELL_CB(), stream.D_LENGTH_CB(), stream.D_EOF_CB(), wr, stream.D_META_CB(), er, None) == stream.DEC_INIT_INVALID_CALLBACKS
    assert d.set_md5_checking(1) and d.get_md5_checking() == 1
    assert d.set_metadata_respond(4) and d.set_metadata_respond(0) and d.set_metadata_respond_all() and not d.set_metadata_respond(127)
    assert d.set_metadata_ignore_application(b"abcd") and d.set_metadata_ignore(2) and d.set_metadata_respond_application(b"abcd")
    assert not d.process_single() and d.finish()
    if engine.device_count() == 0:
        assert d.init() == stream.DEC_INIT_MEMORY_ALLOCATION_ERROR and d.get_state() == stream.DEC_UNINITIALIZED
    d.close()
