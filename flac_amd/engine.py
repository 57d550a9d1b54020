"""ctypes binding of libflacamd.so (include/flac_amd_engine.h).

The library is the product: HIP kernels for gfx950 behind a plain C ABI.  This module only
marshals arguments; it contains no arithmetic of its own and no CPU fallback - if the shared
library or a HIP device is missing the calls raise.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FLACAMD_LIB") or os.path.join(_HERE, "libflacamd.so")   # the override is a development aid for comparing builds

PCM_S32, PCM_S16, PCM_S32_FRAMES = 0, 1, 2

WINDOWS = ["bartlett", "bartlett_hann", "blackman", "blackman_harris_4term_92db", "connes", "flattop", "gauss",
           "hamming", "hann", "kaiser_bessel", "nuttall", "rectangle", "triangle", "tukey", "welch"]

ERRORS = {-1: "invalid argument", -2: "outside the engine limits", -3: "no HIP device", -4: "out of device memory",
          -5: "HIP call failed", -6: "capacity too small", -7: "frame exceeded its slot", -8: "bad stream"}


class Params(C.Structure):
    """flacamd_params"""
    _fields_ = [(n, C.c_uint32) for n in (
        "channels", "bits_per_sample", "sample_rate", "blocksize", "do_mid_side", "loose_mid_side",
        "max_lpc_order", "qlp_coeff_precision", "do_qlp_coeff_prec_search", "do_exhaustive_model_search",
        "min_residual_partition_order", "max_residual_partition_order",
        "disable_constant", "disable_fixed", "disable_verbatim", "num_apodizations")] + [
        ("apod_type", C.c_uint32 * 32), ("apod_param", C.c_float * 32)]


class Md5Group(C.Structure):
    """flacamd_md5_group"""
    _fields_ = [("d_pcm", C.c_void_p), ("pcm_format", C.c_int), ("channels", C.c_uint32), ("bits_per_sample", C.c_uint32), ("nclips", C.c_uint32),
                ("h_start", C.c_void_p), ("h_len", C.c_void_p), ("d_digests", C.c_void_p)]


def md5_groups_device(device, groups, stream=None):
    """groups: list of (d_pcm_ptr, fmt, channels, bits_per_sample, starts, lengths, d_digests_ptr).  One launch hashes every clip
    of every group (flacamd_md5_groups_device); asynchronous on `stream`."""
    arr = (Md5Group * len(groups))()
    keep = []
    for g, (ptr, fmt, ch, bps, starts, lens, dig) in zip(arr, groups):
        st = np.ascontiguousarray(starts, dtype=np.uint64)
        ln = np.ascontiguousarray(lens, dtype=np.uint64)
        keep += [st, ln]
        g.d_pcm, g.pcm_format, g.channels, g.bits_per_sample, g.nclips = ptr, fmt, ch, bps, len(st)
        g.h_start, g.h_len, g.d_digests = st.ctypes.data, ln.ctypes.data, dig
    _check(lib().flacamd_md5_groups_device(device, len(groups), arr, C.c_void_p(stream)), "flacamd_md5_groups_device")


class EngineError(RuntimeError):
    def __init__(self, code, where, detail=""):
        self.code = code
        super().__init__("%s failed: %s (%d)%s" % (where, ERRORS.get(code, "?"), code, (" - " + detail) if detail else ""))


_lib = None


def lib():
    """Load libflacamd.so; raises if it has not been built (python -c 'import __graft_entry__ as g; g.build()')."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FileNotFoundError("%s not built; run __graft_entry__.build()" % LIB_PATH)
        # PyTorch-ROCm bundles its own HIP runtime; two runtimes in one process fight over the
        # device, so when torch is installed it is imported first and libflacamd.so then binds
        # to the runtime torch already loaded (torch provides the device buffers and streams).
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        vp, u32, u64, sz = C.c_void_p, C.c_uint32, C.c_uint64, C.c_size_t
        L.flacamd_params_preset.argtypes = [C.POINTER(Params), C.c_uint]
        L.flacamd_params_preset.restype = None
        L.flacamd_params_resolve.argtypes = [C.POINTER(Params)]
        L.flacamd_window.argtypes = [u32, C.c_float, C.c_int32, vp]
        L.flacamd_window.restype = None
        L.flacamd_encoder_create.argtypes = [C.POINTER(vp), C.c_int, C.POINTER(Params), u32]
        L.flacamd_encoder_destroy.argtypes = [vp]
        L.flacamd_encoder_destroy.restype = None
        L.flacamd_encoder_params.argtypes = [vp]
        L.flacamd_encoder_params.restype = C.POINTER(Params)
        L.flacamd_encoder_max_output_bytes.argtypes = [vp, u32]
        L.flacamd_encoder_max_output_bytes.restype = sz
        L.flacamd_encode_device.argtypes = [vp, vp, C.c_int, u64, C.c_int, u32, u32, vp, sz, vp, C.POINTER(u64), vp]
        L.flacamd_encode_device_queue.argtypes = [vp, vp, C.c_int, u64, C.c_int, u32, u32, vp, sz, vp, vp]
        L.flacamd_encoder_wait.argtypes = [vp, C.POINTER(u64), u32, C.POINTER(u32)]
        L.flacamd_encode_streams_device.argtypes = [vp, vp, C.c_int, u32, vp, vp, vp, sz, vp, C.POINTER(u64), vp]
        L.flacamd_encode_clips_device.argtypes = [vp, vp, C.c_int, u32, vp, vp, vp, sz, vp, C.POINTER(u64), vp]
        L.flacamd_encode_host.argtypes = [vp, vp, C.c_int, u64, C.c_int, u32, u32, vp, sz, vp, C.POINTER(u64), C.POINTER(u32)]
        L.flacamd_encoder_debug_read.argtypes = [vp, C.c_int, vp, sz]
        L.flacamd_encoder_debug_read.restype = C.c_long
        L.flacamd_encoder_spotcheck.argtypes = [vp, C.POINTER(u64 * 5), C.c_int]
        L.flacamd_encoder_set_timing.argtypes = [vp, C.c_int]
        L.flacamd_encoder_set_timing.restype = None
        L.flacamd_encoder_last_timing.argtypes = [vp, C.POINTER(C.c_float * 6)]
        L.flacamd_md5_device.argtypes = [C.c_int, vp, C.c_int, u32, u32, vp, u32, vp, vp]
        L.flacamd_md5_device_static.argtypes = [C.c_int, vp, C.c_int, u32, u32, vp, u32, vp, vp]
        L.flacamd_md5_device_static_max_streams.restype = u32
        L.flacamd_pipe_create.argtypes = [C.POINTER(vp), C.POINTER(C.c_int), u32, C.POINTER(Params), u32, u32]
        L.flacamd_pipe_destroy.argtypes = [vp]
        L.flacamd_pipe_destroy.restype = None
        L.flacamd_pipe_batch_frames.argtypes = [vp]
        L.flacamd_pipe_batch_frames.restype = u32
        L.flacamd_pipe_max_batch_bytes.argtypes = [vp]
        L.flacamd_pipe_max_batch_bytes.restype = sz
        L.flacamd_pipe_params.argtypes = [vp]
        L.flacamd_pipe_params.restype = C.POINTER(Params)
        L.flacamd_pipe_encode_host.argtypes = [vp, vp, C.c_int, u64, C.c_int, u32, vp, sz, vp, sz, C.POINTER(u64), C.POINTER(u32)]
        L.flacamd_decoder_create.argtypes = [C.POINTER(vp), C.c_int, u32, u32, u32, u32, sz]
        L.flacamd_decoder_destroy.argtypes = [vp]
        L.flacamd_decoder_destroy.restype = None
        L.flacamd_decode_device.argtypes = [vp, vp, sz, vp, C.c_int, u64, vp, vp]
        L.flacamd_decode_host.argtypes = [vp, vp, sz, vp, C.c_int, u64, vp]
        L.flacamd_write_stream_head.argtypes = [vp, u32, u32, u32, u32, u32, u32, u64, vp]
        L.flacamd_write_stream_head.restype = sz
        L.flacamd_decoder_frame_table.argtypes = [vp, u32, vp, vp, vp, vp, vp, vp]
        L.flacamd_decoder_set_tolerant.argtypes = [vp, C.c_int]
        L.flacamd_decoder_damage.argtypes = [vp]
        L.flacamd_decoder_damage.restype = u32
        L.flacamd_decoder_candidate_table.argtypes = [vp, u32, C.POINTER(u32), vp, vp, vp, vp, vp, vp]
        L.flacamd_strerror.argtypes = [C.c_int]
        L.flacamd_strerror.restype = C.c_char_p
        L.flacamd_last_error.restype = C.c_char_p
        L.flacamd_device_count.restype = C.c_int
        _lib = L
    return _lib


def _check(rc, where):
    if rc != 0:
        raise EngineError(rc, where, lib().flacamd_last_error().decode(errors="replace"))


def make_params(channels, bits_per_sample, sample_rate, level=8, blocksize=0, apodization=None, **over):
    """Build a Params for a compression level (the reference's -0..-8) with field overrides.
    apodization: list of (window name, parameter)."""
    p = Params()
    p.channels, p.bits_per_sample, p.sample_rate, p.blocksize = channels, bits_per_sample, sample_rate, blocksize
    lib().flacamd_params_preset(C.byref(p), level)
    for k, v in over.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    if apodization:
        p.num_apodizations = len(apodization)
        for i, (name, prm) in enumerate(apodization):
            p.apod_type[i] = WINDOWS.index(name)
            p.apod_param[i] = prm
    return p


def window(name, param, length):
    w = np.empty(length, dtype=np.float32)
    lib().flacamd_window(WINDOWS.index(name), param, length, w.ctypes.data)
    return w


class Encoder:
    """One stream's encoder on one GPU (flacamd_encoder)."""

    def __init__(self, params, max_frames, device=0):
        self._h = C.c_void_p()
        self._lib = lib()
        _check(self._lib.flacamd_encoder_create(C.byref(self._h), device, C.byref(params), max_frames), "flacamd_encoder_create")
        self.params = self._lib.flacamd_encoder_params(self._h).contents
        self.blocksize = int(self.params.blocksize)
        self.channels = int(self.params.channels)
        self.max_frames = max_frames
        self.device = device

    def close(self):
        if self._h:
            self._lib.flacamd_encoder_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def max_output_bytes(self, nframes):
        return int(self._lib.flacamd_encoder_max_output_bytes(self._h, nframes))

    def set_timing(self, on=True):
        self._lib.flacamd_encoder_set_timing(self._h, 1 if on else 0)

    def last_timing(self):
        ms = (C.c_float * 6)()
        _check(self._lib.flacamd_encoder_last_timing(self._h, C.byref(ms)), "flacamd_encoder_last_timing")
        # the autocorrelation and prep run side by side on two streams: the first interval ends when the autocorrelation
        # (on the caller's stream) does, the second is what was left of prep to wait for at the join
        return dict(zip(("autoc_beside_prep", "prep_join_wait", "model", "search", "pack", "gather"), [float(v) for v in ms]))

    def encode_host(self, pcm, final=True, first_frame=0, prev_assignment=0):
        """pcm: numpy (samples, channels) int32 or int16. Returns (list of frame bytes, last assignment)."""
        pcm = np.ascontiguousarray(pcm)
        if pcm.dtype == np.int16:
            fmt = PCM_S16
        elif pcm.dtype == np.int32:
            fmt = PCM_S32
        else:
            raise TypeError("pcm must be int16 or int32")
        n = pcm.shape[0]
        if pcm.ndim != 2 or pcm.shape[1] != self.channels:
            raise ValueError("pcm shape")
        F = (n + self.blocksize - 1) // self.blocksize
        cap = self.max_output_bytes(F)
        out = np.empty(cap, dtype=np.uint8)
        fb = np.zeros(F, dtype=np.uint32)
        total = C.c_uint64(0)
        last = C.c_uint32(0)
        rc = self._lib.flacamd_encode_host(self._h, pcm.ctypes.data, fmt, n, 1 if final else 0, first_frame, prev_assignment,
                                           out.ctypes.data, cap, fb.ctypes.data, C.byref(total), C.byref(last))
        if rc == -6 and total.value > cap:   # frames beyond the usual bound: the call says how many bytes it takes
            cap = int(total.value)
            out = np.empty(cap, dtype=np.uint8)
            rc = self._lib.flacamd_encode_host(self._h, pcm.ctypes.data, fmt, n, 1 if final else 0, first_frame, prev_assignment,
                                               out.ctypes.data, cap, fb.ctypes.data, C.byref(total), C.byref(last))
        _check(rc, "flacamd_encode_host")
        frames, off = [], 0
        for s in fb:
            frames.append(bytes(out[off:off + int(s)]))
            off += int(s)
        assert off == total.value
        return frames, int(last.value)

    def encode_device(self, d_pcm_ptr, fmt, nsamples, d_out_ptr, out_cap, d_frame_bytes_ptr, final=True, first_frame=0,
                      prev_assignment=0, stream=None):
        """Raw device-pointer entry (ints). Returns total bytes."""
        total = C.c_uint64(0)
        _check(self._lib.flacamd_encode_device(self._h, d_pcm_ptr, fmt, nsamples, 1 if final else 0, first_frame, prev_assignment,
                                                d_out_ptr, out_cap, d_frame_bytes_ptr, C.byref(total), stream), "flacamd_encode_device")
        return int(total.value)

    def encode_device_queue(self, d_pcm_ptr, fmt, nsamples, d_out_ptr, out_cap, d_frame_bytes_ptr, final=True, first_frame=0,
                            prev_assignment=0, stream=None):
        """flacamd_encode_device_queue: the batch is queued, nothing is waited for; wait() collects."""
        _check(self._lib.flacamd_encode_device_queue(self._h, d_pcm_ptr, fmt, nsamples, 1 if final else 0, first_frame, prev_assignment,
                                                      d_out_ptr, out_cap, d_frame_bytes_ptr, stream), "flacamd_encode_device_queue")

    def wait(self):
        """flacamd_encoder_wait: total bytes of every queued batch, in queue order."""
        totals = (C.c_uint64 * 16)()
        n = C.c_uint32(0)
        _check(self._lib.flacamd_encoder_wait(self._h, totals, 16, C.byref(n)), "flacamd_encoder_wait")
        return [int(totals[i]) for i in range(n.value)]

    def encode_streams_device(self, d_pcm_ptr, fmt, starts, lengths, d_out_ptr, out_cap, d_frame_bytes_ptr, stream=None):
        """Several complete streams in one pass (flacamd_encode_streams_device): starts / lengths in wide samples,
        the streams tiling the buffer in whole blocks.  Returns total bytes."""
        st = np.ascontiguousarray(starts, dtype=np.uint64)
        ln = np.ascontiguousarray(lengths, dtype=np.uint64)
        total = C.c_uint64(0)
        rc = self._lib.flacamd_encode_streams_device(self._h, d_pcm_ptr, fmt, len(st), st.ctypes.data, ln.ctypes.data, d_out_ptr, out_cap,
                                                     d_frame_bytes_ptr, C.byref(total), stream)
        if rc == -6 and total.value > out_cap:
            raise OutputTooSmall(int(total.value))   # frames beyond the verbatim bound: call again with that many bytes
        _check(rc, "flacamd_encode_streams_device")
        return int(total.value)

    def use_digests(self, d_digests_ptr, ready_event=None):
        """The next encode_clips_device call takes its clips' MD5 digests from device memory (flacamd_encoder_use_digests);
        ready_event: raw hipEvent_t handle (int) waited for before the stream heads are written."""
        _check(self._lib.flacamd_encoder_use_digests(self._h, C.c_void_p(d_digests_ptr), C.c_void_p(ready_event)), "flacamd_encoder_use_digests")

    def encode_clips_device(self, d_pcm_ptr, fmt, starts, lengths, d_out_ptr, out_cap, stream=None):
        """A batch of clips as complete streams assembled on the device (flacamd_encode_clips_device).  Returns the
        byte offsets of the streams within the output (len(starts) + 1 entries)."""
        st = np.ascontiguousarray(starts, dtype=np.uint64)
        ln = np.ascontiguousarray(lengths, dtype=np.uint64)
        offs = np.zeros(len(st) + 1, dtype=np.uint64)
        total = C.c_uint64(0)
        rc = self._lib.flacamd_encode_clips_device(self._h, d_pcm_ptr, fmt, len(st), st.ctypes.data, ln.ctypes.data, d_out_ptr, out_cap,
                                                   offs.ctypes.data, C.byref(total), stream)
        if rc == -6 and total.value > out_cap:
            raise OutputTooSmall(int(total.value))
        _check(rc, "flacamd_encode_clips_device")
        return offs

    def debug(self, what, dtype, count):
        buf = np.empty(count, dtype=dtype)
        got = self._lib.flacamd_encoder_debug_read(self._h, what, buf.ctypes.data, buf.nbytes)
        if got < 0:
            raise EngineError(got, "flacamd_encoder_debug_read")
        return buf[: got // buf.itemsize]

    def spotcheck(self, reset=False):
        """Totals of the estimate spot check (flacamd_encoder_spotcheck): subframes checked, subframes whose estimate
        differed from the packed size, sum and maximum of the differences in bits, bits packed.  Only for an encoder
        created with the switch 0x400000 in FLACAMD_DEBUG."""
        out = (C.c_uint64 * 5)()
        _check(self._lib.flacamd_encoder_spotcheck(self._h, C.byref(out), 1 if reset else 0), "flacamd_encoder_spotcheck")
        return dict(zip(("subframes", "differing", "sum_abs_bits", "max_abs_bits", "packed_bits"), (int(v) for v in out)))


class Pipe:
    """One stream coded from host memory in batches kept in flight over one or several GPUs (flacamd_pipe)."""

    def __init__(self, params, batch_frames, devices=(0,), depth=2):
        self._h = C.c_void_p()
        self._lib = lib()
        ids = (C.c_int * len(devices))(*devices)
        _check(self._lib.flacamd_pipe_create(C.byref(self._h), ids, len(devices), C.byref(params), batch_frames, depth), "flacamd_pipe_create")
        self.params = self._lib.flacamd_pipe_params(self._h).contents
        self.blocksize = int(self.params.blocksize)
        self.channels = int(self.params.channels)
        self.batch_frames = int(self._lib.flacamd_pipe_batch_frames(self._h))

    def close(self):
        if self._h:
            self._lib.flacamd_pipe_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def encode_host(self, pcm, final=True, first_frame=0, out=None):
        """pcm: numpy (samples, channels) int16 or int32 in ordinary host memory.  Returns (frame bytes as one
        uint8 array, sizes) - the frames back to back."""
        pcm = np.ascontiguousarray(pcm)
        fmt = PCM_S16 if pcm.dtype == np.int16 else PCM_S32
        n = pcm.shape[0]
        F = (n + self.blocksize - 1) // self.blocksize
        nb = (F + self.batch_frames - 1) // self.batch_frames
        cap = nb * int(self._lib.flacamd_pipe_max_batch_bytes(self._h)) if out is None else out.nbytes
        if out is None:
            out = np.empty(cap, dtype=np.uint8)
        fb = np.zeros(F, dtype=np.uint32)
        total, nf = C.c_uint64(0), C.c_uint32(0)
        rc = self._lib.flacamd_pipe_encode_host(self._h, pcm.ctypes.data, fmt, n, 1 if final else 0, first_frame, out.ctypes.data, cap,
                                                fb.ctypes.data, F, C.byref(total), C.byref(nf))
        if rc == -6 and total.value > cap:
            raise OutputTooSmall(int(total.value))
        _check(rc, "flacamd_pipe_encode_host")
        assert nf.value == F
        return out[:total.value], fb


def device_count():
    return int(lib().flacamd_device_count())


class OutputTooSmall(Exception):
    """The batch holds frames beyond the usual size bound; .needed is the output size in bytes to call again with."""

    def __init__(self, needed):
        super().__init__("output buffer too small: %d bytes needed" % needed)
        self.needed = needed


class Decoder:
    """Frame-parallel decoder for streams of known channels / bits per sample (flacamd_decoder)."""

    def __init__(self, channels, bits_per_sample, max_blocksize, max_frames, max_bytes, device=0):
        self._h = C.c_void_p()
        self._lib = lib()
        _check(self._lib.flacamd_decoder_create(C.byref(self._h), device, channels, bits_per_sample, max_blocksize, max_frames, max_bytes),
               "flacamd_decoder_create")
        self.channels, self.bps = channels, bits_per_sample

    def close(self):
        if self._h:
            self._lib.flacamd_decoder_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def decode_host(self, data, cap_samples, dtype=np.int32):
        """data: bytes of whole frames. Returns (pcm (samples, channels), info dict)."""
        buf = np.frombuffer(data, dtype=np.uint8)
        fmt = PCM_S16 if np.dtype(dtype) == np.int16 else PCM_S32
        pcm = np.zeros((cap_samples, self.channels), dtype=dtype)
        info = (C.c_uint64 * 4)()
        _check(self._lib.flacamd_decode_host(self._h, buf.ctypes.data, len(buf), pcm.ctypes.data, fmt, cap_samples, info), "flacamd_decode_host")
        return pcm[:int(info[1])], dict(frames=int(info[0]), samples=int(info[1]), first_number=int(info[2]), bytes=int(info[3]))

    def decode_host_frames(self, data, cap_samples):
        """The same with FLACAMD_PCM_S32_FRAMES: returns (flat int32 array, info dict); a frame of n samples that starts at
        wide sample s lies at [s * channels, (s + n) * channels) as channel 0's n samples, then channel 1's, ..."""
        buf = np.frombuffer(data, dtype=np.uint8)
        pcm = np.zeros(cap_samples * self.channels, dtype=np.int32)
        info = (C.c_uint64 * 4)()
        _check(self._lib.flacamd_decode_host(self._h, buf.ctypes.data, len(buf), pcm.ctypes.data, PCM_S32_FRAMES, cap_samples, info), "flacamd_decode_host")
        return pcm[:int(info[1]) * self.channels], dict(frames=int(info[0]), samples=int(info[1]), first_number=int(info[2]), bytes=int(info[3]))

    def set_tolerant(self, on=True):
        """Accept byte ranges whose CRC-valid frames do not tile them (flacamd_decoder_set_tolerant)."""
        _check(self._lib.flacamd_decoder_set_tolerant(self._h, 1 if on else 0), "flacamd_decoder_set_tolerant")

    def damage(self):
        return int(self._lib.flacamd_decoder_damage(self._h))

    def candidate_table(self, max_entries):
        """Every plausible frame start of the last decode: dict of arrays pos, end, info, blocksize, number, sample_off."""
        n = C.c_uint32(0)
        pos, end, num, off = (np.zeros(max_entries, np.uint64) for _ in range(4))
        info, bs = (np.zeros(max_entries, np.uint32) for _ in range(2))
        _check(self._lib.flacamd_decoder_candidate_table(self._h, max_entries, C.byref(n), pos.ctypes.data, end.ctypes.data, info.ctypes.data,
                                                         bs.ctypes.data, num.ctypes.data, off.ctypes.data), "flacamd_decoder_candidate_table")
        k = n.value
        return dict(pos=pos[:k], end=end[:k], info=info[:k], blocksize=bs[:k], number=num[:k], sample_off=off[:k])

    def decode_range(self, data, start, end, slack, cap_samples, max_entries):
        """The frames of `data` (frames only, no metadata) that START in [start, end): the share of one rank when a
        stream is cut into byte ranges.  The window decoded is [start, end + slack); slack >= the longest frame, so
        the last frame of the share is whole.  The first CRC-valid frame at or after `start` opens the chain, every
        further frame starts where the one before ended.  Returns dict(first_pos, end_pos, frames, pcm): absolute
        byte offsets of the share, the number of frames and their samples (samples, channels)."""
        lo, hi = start, min(len(data), end + slack)
        if lo >= hi:
            return dict(first_pos=start, end_pos=start, frames=0, pcm=np.zeros((0, self.channels), np.int32))
        self.set_tolerant(True)
        pcm, _ = self.decode_host(data[lo:hi], cap_samples)
        t = self.candidate_table(max_entries)
        valid = (t["info"] & 3) == 3
        pos, fin, off, bs = (t[k][valid].astype(np.int64) for k in ("pos", "end", "sample_off", "blocksize"))
        first = None
        frames, nxt = 0, None
        for i in range(len(pos)):
            if first is None:
                if pos[i] >= end - lo:
                    break
                first, nxt, s0 = i, pos[i], off[i]
            if pos[i] != nxt:
                continue            # a position inside an accepted frame that happens to check out: not part of the chain
            if pos[i] >= end - lo:
                break
            frames += 1
            nxt = fin[i]
            s1 = off[i] + bs[i]
        if first is None:
            return dict(first_pos=min(end, len(data)), end_pos=min(end, len(data)), frames=0, pcm=np.zeros((0, self.channels), np.int32))
        return dict(first_pos=lo + int(pos[first]), end_pos=lo + int(nxt), frames=frames, pcm=pcm[int(s0):int(s1)].copy())

    def decode_device(self, d_bytes_ptr, nbytes, d_pcm_ptr, fmt, cap_samples, stream=None):
        info = (C.c_uint64 * 4)()
        _check(self._lib.flacamd_decode_device(self._h, d_bytes_ptr, nbytes, d_pcm_ptr, fmt, cap_samples, info, stream), "flacamd_decode_device")
        return dict(frames=int(info[0]), samples=int(info[1]), first_number=int(info[2]), bytes=int(info[3]))


def md5_device(d_pcm_ptr, fmt, channels, bits_per_sample, offsets, device=0, stream=None):
    """MD5 digests of streams held back to back in device memory; offsets = wide-sample offsets (len n+1)."""
    off = np.ascontiguousarray(offsets, dtype=np.uint64)
    n = len(off) - 1
    out = np.zeros((n, 16), dtype=np.uint8)
    _check(lib().flacamd_md5_device(device, d_pcm_ptr, fmt, channels, bits_per_sample, off.ctypes.data, n, out.ctypes.data, stream), "flacamd_md5_device")
    return [bytes(r) for r in out]


def stream_head(blocksize, min_framesize, max_framesize, sample_rate, channels, bits_per_sample, total_samples, md5):
    """'fLaC' + STREAMINFO + VORBIS_COMMENT of a finished stream (flacamd_write_stream_head)."""
    out = np.zeros(86, dtype=np.uint8)
    m = np.frombuffer(bytes(md5), dtype=np.uint8).copy()
    n = lib().flacamd_write_stream_head(out.ctypes.data, blocksize, min_framesize, max_framesize, sample_rate, channels, bits_per_sample, total_samples, m.ctypes.data)
    return bytes(out[:n])
