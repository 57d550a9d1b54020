"""ctypes bindings of the CHECKER libraries (oracle/liboracle.so = this repo's CPU
restatement; oracle/_ref/libflacref.so = the real reference libFLAC + driver).
Test infrastructure only: nothing under flac_amd/ imports this module."""
import ctypes as C
import os
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_SO = os.path.join(ROOT, "oracle", "liboracle.so")
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libflacref.so")
REF_FLAC = os.path.join(ROOT, "oracle", "_ref", "flac")

WINDOWS = ["bartlett", "bartlett_hann", "blackman", "blackman_harris_4term_92db", "connes", "flattop", "gauss",
           "hamming", "hann", "kaiser_bessel", "nuttall", "rectangle", "triangle", "tukey", "welch"]


class FoParams(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in (
        "channels", "bits_per_sample", "sample_rate", "blocksize", "do_mid_side", "loose_mid_side",
        "max_lpc_order", "qlp_coeff_precision", "do_qlp_coeff_prec_search", "do_exhaustive_model_search",
        "min_residual_partition_order", "max_residual_partition_order",
        "disable_constant", "disable_fixed", "disable_verbatim", "num_apodizations")] + [
        ("apod_type", C.c_uint32 * 32), ("apod_param", C.c_float * 32)]


class FoSubInfo(C.Structure):
    _fields_ = [("type", C.c_uint32), ("order", C.c_uint32), ("wasted_bits", C.c_uint32), ("bps", C.c_uint32),
                ("qlp_precision", C.c_uint32), ("qlp_shift", C.c_int32), ("qlp", C.c_int32 * 32),
                ("partition_order", C.c_uint32), ("rice2", C.c_uint32), ("estimate_bits", C.c_uint32),
                ("rice_param", C.c_uint32 * 64)]


class FoFrameInfo(C.Structure):
    _fields_ = [("blocksize", C.c_uint32), ("channel_assignment", C.c_uint32), ("frame_number", C.c_uint32),
                ("bytes", C.c_uint32), ("sub", FoSubInfo * 8), ("autoc", (C.c_float * 33) * 4),
                ("fixed_err", (C.c_uint64 * 5) * 4)]


class RefParams(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in (
        "channels", "bits_per_sample", "sample_rate", "blocksize", "do_mid_side", "loose_mid_side",
        "max_lpc_order", "qlp_coeff_precision", "do_qlp_coeff_prec_search", "do_exhaustive_model_search",
        "min_residual_partition_order", "max_residual_partition_order",
        "disable_constant", "disable_fixed", "disable_verbatim", "streamable_subset")] + [("apodization", C.c_char_p)]


_oracle = None
_ref = None


def oracle_lib():
    global _oracle
    if _oracle is None:
        lib = C.CDLL(ORACLE_SO)
        lib.fo_params_preset.argtypes = [C.POINTER(FoParams), C.c_uint]
        lib.fo_params_resolve.argtypes = [C.POINTER(FoParams)]
        lib.fo_params_resolve.restype = C.c_int
        lib.fo_encode_stream.restype = C.c_long
        lib.fo_encode_stream.argtypes = [C.POINTER(FoParams), C.c_void_p, C.c_uint64, C.c_void_p, C.c_size_t,
                                         C.POINTER(C.c_size_t), C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
        lib.fo_encode_stream_ex.restype = C.c_long
        lib.fo_encode_stream_ex.argtypes = [C.POINTER(FoParams), C.c_void_p, C.c_uint64, C.c_uint32, C.c_int, C.c_void_p, C.c_size_t,
                                            C.POINTER(C.c_size_t), C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
        lib.fo_utf8.argtypes = [C.c_uint32, C.c_void_p, C.POINTER(C.c_uint32)]
        lib.fo_decode_frames.restype = C.c_long
        lib.fo_decode_frames.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint64,
                                         C.POINTER(C.c_uint64), C.c_void_p, C.c_uint32]
        lib.fo_window.argtypes = [C.c_uint32, C.c_float, C.c_int32, C.c_void_p]
        lib.fo_md5_pcm.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p]
        lib.fo_crc8.restype = C.c_uint8
        lib.fo_crc8.argtypes = [C.c_void_p, C.c_size_t]
        lib.fo_crc16.restype = C.c_uint16
        lib.fo_crc16.argtypes = [C.c_void_p, C.c_size_t]
        lib.fo_write_stream_head.argtypes = [C.c_void_p] + [C.c_uint32] * 7 + [C.c_uint64, C.c_void_p, C.c_int]
        _oracle = lib
    return _oracle


def have_ref():
    return os.path.exists(REF_SO)


def ref_lib():
    global _ref
    if _ref is None:
        lib = C.CDLL(REF_SO)
        lib.refdrv_encode.restype = C.c_int
        lib.refdrv_encode.argtypes = [C.POINTER(RefParams), C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_size_t,
                                      C.POINTER(C.c_size_t), C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
        lib.refdrv_decode.restype = C.c_int
        lib.refdrv_decode.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_uint64, C.c_void_p]
        _ref = lib
    return _ref


def make_params(channels, bps, rate, level=8, blocksize=0, apodization=None, **over):
    """Build the (FoParams, RefParams) pair for one configuration.
    apodization: list of (name, param) or None for the preset default."""
    lib = oracle_lib()
    fp = FoParams()
    fp.channels, fp.bits_per_sample, fp.sample_rate, fp.blocksize = channels, bps, rate, blocksize
    lib.fo_params_preset(C.byref(fp), level)
    for k, v in over.items():
        setattr(fp, k, v)
    spec = None
    if apodization:
        fp.num_apodizations = len(apodization)
        parts = []
        for i, (name, prm) in enumerate(apodization):
            fp.apod_type[i] = WINDOWS.index(name)
            fp.apod_param[i] = prm
            parts.append("%s(%g)" % (name, prm) if name in ("tukey", "gauss") else name)
        spec = ";".join(parts).encode()
    rp = RefParams()
    for name, _ in RefParams._fields_[:15]:
        setattr(rp, name, getattr(fp, name))
    rp.streamable_subset = 0
    rp.apodization = spec
    return fp, rp


def _split(buf, sizes):
    out, off = [], 0
    for s in sizes:
        out.append(bytes(buf[off:off + s]))
        off += s
    return out


def oracle_encode(fp, pcm, want_info=False, first_frame=0, final=True):
    """pcm: int32 array (samples, channels). Returns dict(frames=[bytes], samples=[...], md5=bytes, infos=[...])."""
    lib = oracle_lib()
    pcm = np.ascontiguousarray(pcm, dtype=np.int32)
    n, ch = pcm.shape
    bs = fp.blocksize or (4096 if fp.max_lpc_order else 1152)
    maxf = n // bs + 2
    cap = int(n * ch * 4 * 1.1) + maxf * 64 + 4096
    out = np.empty(cap, dtype=np.uint8)
    fb = np.zeros(maxf, dtype=np.uint32)
    fs = np.zeros(maxf, dtype=np.uint32)
    md5 = np.zeros(16, dtype=np.uint8)
    infos = (FoFrameInfo * maxf)() if want_info else None
    outlen = C.c_size_t(0)
    nf = lib.fo_encode_stream_ex(C.byref(fp), pcm.ctypes.data, n, first_frame, 1 if final else 0, out.ctypes.data, cap, C.byref(outlen),
                                 fb.ctypes.data, fs.ctypes.data, maxf, C.cast(infos, C.c_void_p) if want_info else None, md5.ctypes.data)
    if nf < 0:
        raise RuntimeError("fo_encode_stream failed: %d" % nf)
    return dict(frames=_split(out, fb[:nf]), samples=[int(v) for v in fs[:nf]], md5=bytes(md5), infos=infos, nframes=nf)


def ref_encode(rp, pcm, chunk=0):
    lib = ref_lib()
    pcm = np.ascontiguousarray(pcm, dtype=np.int32)
    n, ch = pcm.shape
    bs = rp.blocksize or (4096 if rp.max_lpc_order else 1152)
    maxf = n // bs + 2
    cap = int(n * ch * 4 * 1.1) + maxf * 64 + 65536
    out = np.empty(cap, dtype=np.uint8)
    fb = np.zeros(maxf, dtype=np.uint32)
    fs = np.zeros(maxf, dtype=np.uint32)
    info = np.zeros(5, dtype=np.uint64)
    md5 = np.zeros(16, dtype=np.uint8)
    outlen = C.c_size_t(0)
    rc = lib.refdrv_encode(C.byref(rp), pcm.ctypes.data, n, chunk, out.ctypes.data, cap, C.byref(outlen), fb.ctypes.data,
                           fs.ctypes.data, maxf, info.ctypes.data, md5.ctypes.data)
    if rc != 0:
        raise RuntimeError("refdrv_encode failed: %d" % rc)
    nf = int(info[1])
    audio = int(info[0])
    frames = _split(out[audio:], fb[:nf])
    return dict(frames=frames, samples=[int(v) for v in fs[:nf]], md5=bytes(md5), stream=bytes(out[:outlen.value]),
                audio_offset=audio, min_framesize=int(info[2]), max_framesize=int(info[3]), total_samples=int(info[4]))


def ref_decode(stream, cap_samples, channels):
    lib = ref_lib()
    buf = np.frombuffer(stream, dtype=np.uint8)
    pcm = np.zeros((cap_samples, channels), dtype=np.int32)
    info = np.zeros(6, dtype=np.uint64)
    rc = lib.refdrv_decode(buf.ctypes.data, len(buf), pcm.ctypes.data, cap_samples, info.ctypes.data)
    return rc, pcm[:int(info[0])], [int(v) for v in info]


def oracle_decode(frames_bytes, channels, bps, cap_samples):
    lib = oracle_lib()
    buf = np.frombuffer(frames_bytes, dtype=np.uint8)
    pcm = np.zeros((cap_samples, channels), dtype=np.int32)
    ns = C.c_uint64(0)
    nf = lib.fo_decode_frames(buf.ctypes.data, len(buf), channels, bps, pcm.ctypes.data, cap_samples, C.byref(ns), None, 0)
    return nf, pcm[:ns.value]


def stream_head(min_bs, max_bs, min_fs, max_fs, rate, channels, bps, total, md5, is_last=True):
    lib = oracle_lib()
    out = np.zeros(42, dtype=np.uint8)
    m = np.frombuffer(md5, dtype=np.uint8).copy()
    lib.fo_write_stream_head(out.ctypes.data, min_bs, max_bs, min_fs, max_fs, rate, channels, bps, total, m.ctypes.data, 1 if is_last else 0)
    return bytes(out)
