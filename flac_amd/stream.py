"""ctypes binding of the libFLAC-style stream encoder object (include/flac_amd_stream.h).

`StreamEncoder` mirrors the reference's C++ wrapper FLAC::Encoder::Stream / File
(include/FLAC++/encoder.h:101-255): setters before init, virtual write/seek/tell/metadata
callbacks to override, process(), process_interleaved(), finish().  All work happens in
libflacamd.so; this file only marshals.
"""
import ctypes as C

import numpy as np

from . import engine

OK, UNINITIALIZED, OGG_ERROR, VERIFY_DECODER_ERROR, VERIFY_MISMATCH, CLIENT_ERROR, IO_ERROR, FRAMING_ERROR, MEMORY_ALLOCATION_ERROR = range(9)
(INIT_OK, INIT_ENCODER_ERROR, INIT_UNSUPPORTED_CONTAINER, INIT_INVALID_CALLBACKS, INIT_INVALID_NUMBER_OF_CHANNELS,
 INIT_INVALID_BITS_PER_SAMPLE, INIT_INVALID_SAMPLE_RATE, INIT_INVALID_BLOCK_SIZE, INIT_INVALID_MAX_LPC_ORDER,
 INIT_INVALID_QLP_COEFF_PRECISION, INIT_BLOCK_SIZE_TOO_SMALL_FOR_LPC_ORDER, INIT_NOT_STREAMABLE, INIT_INVALID_METADATA,
 INIT_ALREADY_INITIALIZED) = range(14)


class StreamInfo(C.Structure):
    _fields_ = [("min_blocksize", C.c_uint), ("max_blocksize", C.c_uint), ("min_framesize", C.c_uint), ("max_framesize", C.c_uint),
                ("sample_rate", C.c_uint), ("channels", C.c_uint), ("bits_per_sample", C.c_uint), ("total_samples", C.c_uint64),
                ("md5sum", C.c_uint8 * 16)]


class _MetaData(C.Union):
    _fields_ = [("stream_info", StreamInfo), ("reserve_", C.c_uint64 * 32)]


class StreamMetadata(C.Structure):
    _fields_ = [("type", C.c_int), ("is_last", C.c_int), ("length", C.c_uint), ("data", _MetaData)]


WRITE_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint8), C.c_size_t, C.c_uint, C.c_uint, C.c_void_p)
SEEK_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint64, C.c_void_p)
TELL_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint64), C.c_void_p)
META_CB = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(StreamMetadata), C.c_void_p)
PROGRESS_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint, C.c_uint, C.c_void_p)

_SETTERS_UINT = ["verify", "streamable_subset", "channels", "bits_per_sample", "sample_rate", "compression_level", "blocksize",
                 "do_mid_side_stereo", "loose_mid_side_stereo", "max_lpc_order", "qlp_coeff_precision", "do_qlp_coeff_prec_search",
                 "do_escape_coding", "do_exhaustive_model_search", "min_residual_partition_order", "max_residual_partition_order",
                 "rice_parameter_search_dist", "do_md5"]
_DISABLERS = ["disable_constant_subframes", "disable_fixed_subframes", "disable_verbatim_subframes"]
_GETTERS = ["state", "verify", "streamable_subset", "channels", "bits_per_sample", "sample_rate", "blocksize", "do_mid_side_stereo",
            "loose_mid_side_stereo", "max_lpc_order", "qlp_coeff_precision", "do_qlp_coeff_prec_search", "do_escape_coding",
            "do_exhaustive_model_search", "min_residual_partition_order", "max_residual_partition_order", "rice_parameter_search_dist"]

_bound = False


def _lib():
    global _bound
    L = engine.lib()
    if not _bound:
        vp = C.c_void_p
        L.FLAC__stream_encoder_new.restype = vp
        L.FLAC__stream_encoder_delete.argtypes = [vp]
        L.FLAC__stream_encoder_delete.restype = None
        for n in _SETTERS_UINT:
            getattr(L, "FLAC__stream_encoder_set_" + n).argtypes = [vp, C.c_uint]
        for n in _DISABLERS:
            getattr(L, "FLAC__stream_encoder_" + n).argtypes = [vp, C.c_int]
        L.FLAC__stream_encoder_set_apodization.argtypes = [vp, C.c_char_p]
        L.FLAC__stream_encoder_set_total_samples_estimate.argtypes = [vp, C.c_uint64]
        L.FLAC__stream_encoder_set_metadata.argtypes = [vp, vp, C.c_uint]
        for n in _GETTERS:
            f = getattr(L, "FLAC__stream_encoder_get_" + n)
            f.argtypes, f.restype = [vp], C.c_uint
        L.FLAC__stream_encoder_get_total_samples_estimate.argtypes = [vp]
        L.FLAC__stream_encoder_get_total_samples_estimate.restype = C.c_uint64
        L.FLAC__stream_encoder_get_resolved_state_string.argtypes = [vp]
        L.FLAC__stream_encoder_get_resolved_state_string.restype = C.c_char_p
        L.FLAC__stream_encoder_init_stream.argtypes = [vp, WRITE_CB, SEEK_CB, TELL_CB, META_CB, vp]
        L.FLAC__stream_encoder_init_file.argtypes = [vp, C.c_char_p, PROGRESS_CB, vp]
        L.FLAC__stream_encoder_process.argtypes = [vp, C.POINTER(C.c_void_p), C.c_uint]
        L.FLAC__stream_encoder_process_interleaved.argtypes = [vp, vp, C.c_uint]
        L.FLAC__stream_encoder_finish.argtypes = [vp]
        L.flacamd_stream_encoder_set_device.argtypes = [vp, C.c_int]
        L.flacamd_stream_encoder_set_batch_frames.argtypes = [vp, C.c_uint]
        _bound = True
    return L


class StreamEncoder:
    """FLAC::Encoder::Stream: subclass and override write_callback (and optionally
    seek_callback / tell_callback / metadata_callback), or use `File`."""

    seekable = False

    def __init__(self):
        self._lib = _lib()
        self._h = C.c_void_p(self._lib.FLAC__stream_encoder_new())
        if not self._h:
            raise MemoryError("FLAC__stream_encoder_new")
        self._keep = None

    def close(self):
        if self._h:
            self._lib.FLAC__stream_encoder_delete(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __getattr__(self, name):
        # set_xxx / get_xxx / disable_xxx forward to the C functions of the same name
        if name.startswith(("set_", "get_", "disable_")):
            fn = getattr(self._lib, ("FLAC__stream_encoder_" if not name.endswith(("set_device", "set_batch_frames")) else "flacamd_stream_encoder_") + name)
            if name.startswith("get_"):
                return lambda: fn(self._h)
            if name == "set_apodization":
                return lambda spec: bool(fn(self._h, spec.encode() if isinstance(spec, str) else spec))
            return lambda value: bool(fn(self._h, int(value)))
        raise AttributeError(name)

    def get_resolved_state_string(self):
        return self._lib.FLAC__stream_encoder_get_resolved_state_string(self._h).decode()

    # callbacks to override -------------------------------------------------------------
    def write_callback(self, data, samples, current_frame):
        raise NotImplementedError

    def seek_callback(self, offset):
        return 2  # FLAC__STREAM_ENCODER_SEEK_STATUS_UNSUPPORTED

    def tell_callback(self):
        return None

    def metadata_callback(self, streaminfo):
        pass

    def init(self):
        def w(_e, buf, nbytes, samples, frame, _cd):
            try:
                return int(self.write_callback(C.string_at(buf, nbytes), samples, frame) or 0)
            except Exception:
                return 1

        def s(_e, off, _cd):
            return int(self.seek_callback(off) or 0)

        def t(_e, poff, _cd):
            v = self.tell_callback()
            if v is None:
                return 2
            poff[0] = v
            return 0

        def m(_e, md, _cd):
            si = md.contents.data.stream_info
            self.metadata_callback(dict(min_blocksize=si.min_blocksize, max_blocksize=si.max_blocksize, min_framesize=si.min_framesize,
                                        max_framesize=si.max_framesize, sample_rate=si.sample_rate, channels=si.channels,
                                        bits_per_sample=si.bits_per_sample, total_samples=si.total_samples, md5sum=bytes(si.md5sum)))

        cbs = (WRITE_CB(w), SEEK_CB(s) if self.seekable else SEEK_CB(), TELL_CB(t) if self.seekable else TELL_CB(), META_CB(m))
        status = int(self._lib.FLAC__stream_encoder_init_stream(self._h, *cbs, None))
        if status == INIT_OK:
            self._keep = cbs  # the C object now points at these thunks; a refused init leaves the previous ones in use
        return status

    def process_interleaved(self, pcm):
        """pcm: int32 array (samples, channels)"""
        pcm = np.ascontiguousarray(pcm, dtype=np.int32)
        return bool(self._lib.FLAC__stream_encoder_process_interleaved(self._h, pcm.ctypes.data, pcm.shape[0]))

    def process(self, channels):
        """channels: list of int32 arrays, one per channel"""
        arrs = [np.ascontiguousarray(c, dtype=np.int32) for c in channels]
        ptrs = (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
        return bool(self._lib.FLAC__stream_encoder_process(self._h, ptrs, len(arrs[0])))

    def finish(self):
        return bool(self._lib.FLAC__stream_encoder_finish(self._h))


class MemoryEncoder(StreamEncoder):
    """Seekable in-memory sink: collects the stream in a bytearray and records every write."""
    seekable = True

    def __init__(self, seekable=True):
        super().__init__()
        self.seekable = seekable
        self.buf = bytearray()
        self.pos = 0
        self.writes = []   # (bytes, samples, current_frame)
        self.streaminfo = None

    def write_callback(self, data, samples, current_frame):
        self.writes.append((len(data), samples, current_frame))
        self.buf[self.pos:self.pos + len(data)] = data
        self.pos += len(data)
        return 0

    def seek_callback(self, offset):
        self.pos = offset
        return 0

    def tell_callback(self):
        return self.pos

    def metadata_callback(self, streaminfo):
        self.streaminfo = streaminfo


class FileEncoder(StreamEncoder):
    """FLAC::Encoder::File (include/FLAC++/encoder.h:222-255)."""

    def __init__(self):
        super().__init__()
        self.progress = []

    def init(self, filename):
        def p(_e, nbytes, nsamples, nframes, total_est, _cd):
            self.progress.append((nbytes, nsamples, nframes, total_est))

        cb = PROGRESS_CB(p)
        status = int(self._lib.FLAC__stream_encoder_init_file(self._h, filename.encode(), cb, None))
        if status == INIT_OK:
            self._keep = (cb,)
        return status
