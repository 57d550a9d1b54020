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


class SeekPoint(C.Structure):
    _fields_ = [("sample_number", C.c_uint64), ("stream_offset", C.c_uint64), ("frame_samples", C.c_uint)]


class SeekTable(C.Structure):
    _fields_ = [("num_points", C.c_uint), ("points", C.POINTER(SeekPoint))]


class Application(C.Structure):
    _fields_ = [("id", C.c_uint8 * 4), ("data", C.POINTER(C.c_uint8))]


class VorbisCommentEntry(C.Structure):
    _fields_ = [("length", C.c_uint32), ("entry", C.POINTER(C.c_uint8))]


class VorbisComment(C.Structure):
    _fields_ = [("vendor_string", VorbisCommentEntry), ("num_comments", C.c_uint32), ("comments", C.POINTER(VorbisCommentEntry))]


class CueSheetIndex(C.Structure):
    _fields_ = [("offset", C.c_uint64), ("number", C.c_uint8)]


class CueSheetTrack(C.Structure):
    _fields_ = [("offset", C.c_uint64), ("number", C.c_uint8), ("isrc", C.c_char * 13), ("flags", C.c_uint8),   # bit 0: type (1 = non-audio), bit 1: pre_emphasis
                ("num_indices", C.c_uint8), ("indices", C.POINTER(CueSheetIndex))]


class CueSheet(C.Structure):
    _fields_ = [("media_catalog_number", C.c_char * 129), ("lead_in", C.c_uint64), ("is_cd", C.c_int), ("num_tracks", C.c_uint),
                ("tracks", C.POINTER(CueSheetTrack))]


class Picture(C.Structure):
    _fields_ = [("type", C.c_int), ("mime_type", C.c_char_p), ("description", C.c_char_p), ("width", C.c_uint32), ("height", C.c_uint32),
                ("depth", C.c_uint32), ("colors", C.c_uint32), ("data_length", C.c_uint32), ("data", C.POINTER(C.c_uint8))]


class Unknown(C.Structure):
    _fields_ = [("data", C.POINTER(C.c_uint8))]


class _MetaData(C.Union):
    _fields_ = [("stream_info", StreamInfo), ("padding", C.c_int), ("application", Application), ("seek_table", SeekTable),
                ("vorbis_comment", VorbisComment), ("cue_sheet", CueSheet), ("picture", Picture), ("unknown", Unknown)]


class StreamMetadata(C.Structure):
    """FLAC__StreamMetadata (include/flac_amd_stream.h).  Blocks built by the helpers below keep their
    buffers alive in `_keep`; like in C they must outlive the encoder's finish()."""
    _fields_ = [("type", C.c_int), ("is_last", C.c_int), ("length", C.c_uint), ("data", _MetaData)]


(META_STREAMINFO, META_PADDING, META_APPLICATION, META_SEEKTABLE, META_VORBIS_COMMENT, META_CUESHEET, META_PICTURE) = range(7)
SEEKPOINT_PLACEHOLDER = 0xFFFFFFFFFFFFFFFF


def _bytes_ptr(b):
    buf = (C.c_uint8 * max(1, len(b))).from_buffer_copy(bytes(b) + (b"" if len(b) else b"\0"))
    return buf, C.cast(buf, C.POINTER(C.c_uint8))


def padding_block(length):
    m = StreamMetadata(type=META_PADDING, length=length)
    m._keep = []
    return m


def application_block(app_id, data):
    m = StreamMetadata(type=META_APPLICATION, length=4 + len(data))
    buf, ptr = _bytes_ptr(data)
    m.data.application.id[:] = list(app_id)
    m.data.application.data = ptr
    m._keep = [buf]
    return m


def seektable_block(sample_numbers):
    """template: target sample numbers (SEEKPOINT_PLACEHOLDER for placeholders); the encoder fills it in"""
    pts = (SeekPoint * max(1, len(sample_numbers)))()
    for i, n in enumerate(sample_numbers):
        pts[i].sample_number = n
    m = StreamMetadata(type=META_SEEKTABLE, length=18 * len(sample_numbers))
    m.data.seek_table.num_points = len(sample_numbers)
    m.data.seek_table.points = C.cast(pts, C.POINTER(SeekPoint))
    m._keep = [pts]
    return m


def spaced_seekpoints(step_samples, total_samples):
    """target samples of `-S <n>s`: one point every step_samples from 0, none at total_samples
    (metadata_object.c:1108-1139)"""
    num = 1 + total_samples // step_samples - (1 if total_samples % step_samples == 0 else 0)
    return [i * step_samples for i in range(num)]


def vorbis_comment_block(comments):
    ents = (VorbisCommentEntry * max(1, len(comments)))()
    keep = [ents]
    length = 4 + 4
    for i, c in enumerate(comments):
        c = c.encode() if isinstance(c, str) else bytes(c)
        buf, ptr = _bytes_ptr(c)
        keep.append(buf)
        ents[i].length, ents[i].entry = len(c), ptr
        length += 4 + len(c)
    m = StreamMetadata(type=META_VORBIS_COMMENT, length=length)
    m.data.vorbis_comment.num_comments = len(comments)
    m.data.vorbis_comment.comments = C.cast(ents, C.POINTER(VorbisCommentEntry))
    m._keep = keep
    return m


def picture_block(pic_type, mime, description, width, height, depth, colors, data):
    mime_b = mime.encode() if isinstance(mime, str) else mime
    desc_b = description.encode() if isinstance(description, str) else description
    buf, ptr = _bytes_ptr(data)
    m = StreamMetadata(type=META_PICTURE, length=4 + 4 + len(mime_b) + 4 + len(desc_b) + 16 + 4 + len(data))
    p = m.data.picture
    p.type, p.mime_type, p.description = pic_type, mime_b, desc_b
    p.width, p.height, p.depth, p.colors, p.data_length, p.data = width, height, depth, colors, len(data), ptr
    m._keep = [buf, mime_b, desc_b]
    return m


def cuesheet_block(tracks, media_catalog_number=b"", lead_in=88200, is_cd=True):
    """tracks: list of (offset, number, isrc, is_data, pre_emphasis, [(index_offset, index_number), ...]); the last one is the lead-out"""
    trs = (CueSheetTrack * len(tracks))()
    keep = [trs]
    length = (128 * 8 + 64 + 1 + 7 + 258 * 8 + 8) // 8
    for i, (off, num, isrc, is_data, pre, idx) in enumerate(tracks):
        ix = (CueSheetIndex * max(1, len(idx)))()
        for j, (io, inum) in enumerate(idx):
            ix[j].offset, ix[j].number = io, inum
        keep.append(ix)
        trs[i].offset, trs[i].number, trs[i].isrc, trs[i].flags = off, num, isrc, int(bool(is_data)) | (int(bool(pre)) << 1)
        trs[i].num_indices, trs[i].indices = len(idx), C.cast(ix, C.POINTER(CueSheetIndex))
        length += (64 + 8 + 12 * 8 + 1 + 1 + 6 + 13 * 8 + 8) // 8 + len(idx) * (64 + 8 + 3 * 8) // 8
    m = StreamMetadata(type=META_CUESHEET, length=length)
    cs = m.data.cue_sheet
    cs.media_catalog_number, cs.lead_in, cs.is_cd, cs.num_tracks = media_catalog_number, lead_in, int(is_cd), len(tracks)
    cs.tracks = C.cast(trs, C.POINTER(CueSheetTrack))
    m._keep = keep
    return m


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
        L.flacamd_stream_encoder_set_batches_in_flight.argtypes = [vp, C.c_uint]
        L.flacamd_stream_encoder_set_devices.argtypes = [vp, C.c_uint, C.POINTER(C.c_int)]
        L.FLAC__stream_encoder_get_verify_decoder_state.argtypes = [vp]
        L.FLAC__stream_encoder_get_verify_decoder_state.restype = C.c_int
        L.FLAC__stream_encoder_get_verify_decoder_error_stats.argtypes = [vp, C.POINTER(C.c_uint64)] + [C.POINTER(C.c_uint)] * 3 + [C.POINTER(C.c_int32)] * 2
        L.FLAC__stream_encoder_get_verify_decoder_error_stats.restype = None
        L.flacamd_stream_encoder_inject_verify_fault.argtypes = [vp, C.c_uint64, C.c_uint, C.c_int32]
        L.flacamd_stream_encoder_inject_frame_fault.argtypes = [vp, C.c_uint, C.c_uint, C.c_uint]
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
            fn = getattr(self._lib, ("FLAC__stream_encoder_" if not name.endswith(("set_device", "set_batch_frames", "set_batches_in_flight")) else "flacamd_stream_encoder_") + name)
            if name.startswith("get_"):
                return lambda: fn(self._h)
            if name == "set_apodization":
                return lambda spec: bool(fn(self._h, spec.encode() if isinstance(spec, str) else spec))
            return lambda value: bool(fn(self._h, int(value)))
        raise AttributeError(name)

    def get_resolved_state_string(self):
        return self._lib.FLAC__stream_encoder_get_resolved_state_string(self._h).decode()

    def set_devices(self, ids):
        """Batches go to these HIP devices in turn (flacamd_stream_encoder_set_devices)."""
        arr = (C.c_int * len(ids))(*ids)
        return bool(self._lib.flacamd_stream_encoder_set_devices(self._h, len(ids), arr))

    def set_metadata(self, blocks):
        """blocks: list of StreamMetadata (see padding_block() etc.); they are kept alive by this object"""
        self._meta = list(blocks)
        arr = (C.POINTER(StreamMetadata) * max(1, len(blocks)))(*[C.pointer(b) for b in blocks])
        self._meta_arr = arr
        return bool(self._lib.FLAC__stream_encoder_set_metadata(self._h, C.cast(arr, C.c_void_p) if blocks else None, len(blocks)))

    def get_verify_decoder_error_stats(self):
        a = C.c_uint64()
        f, c, sm = C.c_uint(), C.c_uint(), C.c_uint()
        ex, got = C.c_int32(), C.c_int32()
        self._lib.FLAC__stream_encoder_get_verify_decoder_error_stats(self._h, a, f, c, sm, ex, got)
        return dict(absolute_sample=a.value, frame_number=f.value, channel=c.value, sample=sm.value, expected=ex.value, got=got.value)

    def inject_verify_fault(self, absolute_sample, channel, mask):
        return bool(self._lib.flacamd_stream_encoder_inject_verify_fault(self._h, absolute_sample, channel, mask))

    def inject_frame_fault(self, frame_number, byte_index, mask):
        return bool(self._lib.flacamd_stream_encoder_inject_frame_fault(self._h, frame_number, byte_index, mask))

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


def _block_to_dict(b):
    """A delivered metadata block as plain Python data (the C object is only valid during the callback)."""
    d = dict(type=b.type, is_last=bool(b.is_last), length=b.length)
    u = b.data
    if b.type == META_APPLICATION:
        d.update(id=bytes(u.application.id), data=C.string_at(u.application.data, b.length - 4) if b.length > 4 else b"")
    elif b.type == META_SEEKTABLE:
        t = u.seek_table
        d.update(points=[(t.points[i].sample_number, t.points[i].stream_offset, t.points[i].frame_samples) for i in range(t.num_points)])
    elif b.type == META_VORBIS_COMMENT:
        v = u.vorbis_comment
        d.update(vendor=C.string_at(v.vendor_string.entry, v.vendor_string.length) if v.vendor_string.length else b"",
                 comments=[C.string_at(v.comments[i].entry, v.comments[i].length) if v.comments[i].length else b"" for i in range(v.num_comments)])
    elif b.type == META_CUESHEET:
        cs = u.cue_sheet
        tracks = []
        for i in range(cs.num_tracks):
            t = cs.tracks[i]
            tracks.append((t.offset, t.number, t.isrc, bool(t.flags & 1), bool(t.flags & 2), [(t.indices[j].offset, t.indices[j].number) for j in range(t.num_indices)]))
        d.update(media_catalog_number=cs.media_catalog_number, lead_in=cs.lead_in, is_cd=bool(cs.is_cd), tracks=tracks)
    elif b.type == META_PICTURE:
        p = u.picture
        d.update(picture_type=p.type, mime_type=p.mime_type, description=p.description, width=p.width, height=p.height, depth=p.depth,
                 colors=p.colors, data=C.string_at(p.data, p.data_length))
    elif b.type != META_PADDING:
        d.update(data=C.string_at(u.unknown.data, b.length) if b.length else b"")
    return d


# ---- decoder object -----------------------------------------------------------------------

(DEC_SEARCH_FOR_METADATA, DEC_READ_METADATA, DEC_SEARCH_FOR_FRAME_SYNC, DEC_READ_FRAME, DEC_END_OF_STREAM, DEC_OGG_ERROR, DEC_SEEK_ERROR,
 DEC_ABORTED, DEC_MEMORY_ALLOCATION_ERROR, DEC_UNINITIALIZED) = range(10)
(DEC_INIT_OK, DEC_INIT_UNSUPPORTED_CONTAINER, DEC_INIT_INVALID_CALLBACKS, DEC_INIT_MEMORY_ALLOCATION_ERROR, DEC_INIT_ERROR_OPENING_FILE,
 DEC_INIT_ALREADY_INITIALIZED) = range(6)
ERR_LOST_SYNC, ERR_BAD_HEADER, ERR_FRAME_CRC_MISMATCH, ERR_UNPARSEABLE_STREAM = range(4)


class _FrameNumber(C.Union):
    _fields_ = [("frame_number", C.c_uint32), ("sample_number", C.c_uint64)]


class FrameHeader(C.Structure):
    _fields_ = [("blocksize", C.c_uint), ("sample_rate", C.c_uint), ("channels", C.c_uint), ("channel_assignment", C.c_int),
                ("bits_per_sample", C.c_uint), ("number_type", C.c_int), ("number", _FrameNumber), ("crc", C.c_uint8)]


class RiceContents(C.Structure):
    _fields_ = [("parameters", C.POINTER(C.c_uint)), ("raw_bits", C.POINTER(C.c_uint)), ("capacity_by_order", C.c_uint)]


class PartitionedRice(C.Structure):
    _fields_ = [("order", C.c_uint), ("contents", C.POINTER(RiceContents))]


class EntropyCodingMethod(C.Structure):
    _fields_ = [("type", C.c_int), ("partitioned_rice", PartitionedRice)]


class SubframeFixed(C.Structure):
    _fields_ = [("entropy_coding_method", EntropyCodingMethod), ("order", C.c_uint), ("warmup", C.c_int32 * 4), ("residual", C.POINTER(C.c_int32))]


class SubframeLPC(C.Structure):
    _fields_ = [("entropy_coding_method", EntropyCodingMethod), ("order", C.c_uint), ("qlp_coeff_precision", C.c_uint), ("quantization_level", C.c_int),
                ("qlp_coeff", C.c_int32 * 32), ("warmup", C.c_int32 * 32), ("residual", C.POINTER(C.c_int32))]


class _SubframeData(C.Union):
    _fields_ = [("constant", C.c_int32), ("fixed", SubframeFixed), ("lpc", SubframeLPC), ("verbatim", C.POINTER(C.c_int32))]


class Subframe(C.Structure):
    _fields_ = [("type", C.c_int), ("data", _SubframeData), ("wasted_bits", C.c_uint)]


class Frame(C.Structure):
    """FLAC__Frame (include/FLAC/format.h:472-476)"""
    _fields_ = [("header", FrameHeader), ("subframes", Subframe * 8), ("footer_crc", C.c_uint16)]


def subframe_summary(frame):
    """The analysis view of a decoded frame (what src/flac/analyze.c prints): one dict per channel."""
    out = []
    for c in range(frame.header.channels):
        sf = frame.subframes[c]
        d = dict(type=int(sf.type), wasted_bits=int(sf.wasted_bits))
        body = sf.data.fixed if sf.type == 2 else sf.data.lpc if sf.type == 3 else None
        if sf.type == 0:
            d["value"] = int(sf.data.constant)
        if body is not None:
            po = int(body.entropy_coding_method.partitioned_rice.order)
            pc = body.entropy_coding_method.partitioned_rice.contents.contents
            d.update(order=int(body.order), warmup=[int(body.warmup[i]) for i in range(body.order)], method=int(body.entropy_coding_method.type), partition_order=po,
                     parameters=[int(pc.parameters[i]) for i in range(1 << po)], raw_bits=[int(pc.raw_bits[i]) for i in range(1 << po)])
            if sf.type == 3:
                d.update(qlp_coeff_precision=int(body.qlp_coeff_precision), quantization_level=int(body.quantization_level), qlp_coeff=[int(body.qlp_coeff[i]) for i in range(body.order)])
        out.append(d)
    return out


D_READ_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint8), C.POINTER(C.c_size_t), C.c_void_p)
D_SEEK_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint64, C.c_void_p)
D_TELL_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint64), C.c_void_p)
D_LENGTH_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint64), C.c_void_p)
D_EOF_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p)
D_WRITE_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(Frame), C.POINTER(C.POINTER(C.c_int32)), C.c_void_p)
D_META_CB = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(StreamMetadata), C.c_void_p)
D_ERROR_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_void_p)

_dec_bound = False


def bind_decoder_api(L, own=True):
    """ctypes signatures of the FLAC__stream_decoder_* functions on library L (this library, or any other
    that exports the same API); own=False leaves out the flacamd_* additions."""
    vp = C.c_void_p
    if True:
        L.FLAC__stream_decoder_new.restype = vp
        L.FLAC__stream_decoder_delete.argtypes = [vp]
        L.FLAC__stream_decoder_delete.restype = None
        for n in ("set_md5_checking", "set_metadata_respond", "set_metadata_ignore"):
            getattr(L, "FLAC__stream_decoder_" + n).argtypes = [vp, C.c_int]
        for n in ("set_metadata_respond_all", "set_metadata_ignore_all", "finish", "flush", "reset", "process_single",
                  "process_until_end_of_metadata", "process_until_end_of_stream", "skip_single_frame", "get_state", "get_md5_checking",
                  "get_channels", "get_channel_assignment", "get_bits_per_sample", "get_sample_rate", "get_blocksize"):
            getattr(L, "FLAC__stream_decoder_" + n).argtypes = [vp]
        L.FLAC__stream_decoder_get_total_samples.argtypes = [vp]
        L.FLAC__stream_decoder_get_total_samples.restype = C.c_uint64
        L.FLAC__stream_decoder_get_resolved_state_string.argtypes = [vp]
        L.FLAC__stream_decoder_get_resolved_state_string.restype = C.c_char_p
        L.FLAC__stream_decoder_get_decode_position.argtypes = [vp, C.POINTER(C.c_uint64)]
        L.FLAC__stream_decoder_seek_absolute.argtypes = [vp, C.c_uint64]
        L.FLAC__stream_decoder_init_stream.argtypes = [vp, D_READ_CB, D_SEEK_CB, D_TELL_CB, D_LENGTH_CB, D_EOF_CB, D_WRITE_CB, D_META_CB, D_ERROR_CB, vp]
        L.FLAC__stream_decoder_init_file.argtypes = [vp, C.c_char_p, D_WRITE_CB, D_META_CB, D_ERROR_CB, vp]
        if own:
            L.flacamd_stream_decoder_set_device.argtypes = [vp, C.c_int]
            L.flacamd_stream_decoder_set_batch_frames.argtypes = [vp, C.c_uint]
        L.FLAC__stream_decoder_set_metadata_respond_application.argtypes = [vp, C.c_char_p]
        L.FLAC__stream_decoder_set_metadata_ignore_application.argtypes = [vp, C.c_char_p]
    return L


def _dec_lib():
    global _dec_bound
    L = engine.lib()
    if not _dec_bound:
        bind_decoder_api(L)
        _dec_bound = True
    return L


class StreamDecoder:
    """FLAC::Decoder::Stream (include/FLAC++/decoder.h:97-208) over bytes held in memory (or a file):
    collects what the callbacks deliver.  frames: list of (header dict, int32 array (channels, blocksize))."""

    def __init__(self, data=None, seekable=True, read_chunk=None, lib=None):
        self._lib = lib if lib is not None else _dec_lib()   # lib: another library with the same API, already bound
        self._h = C.c_void_p(self._lib.FLAC__stream_decoder_new())
        self.data, self.pos, self.seekable, self.read_chunk = data, 0, seekable, read_chunk
        self.frames, self.errors, self.streaminfo = [], [], None
        self.events = []     # ("error", status) and ("frame", sample_number, blocksize) in the order the callbacks came
        self.metadata = []   # every non-STREAMINFO block delivered, as plain dicts (see _block_to_dict)
        self._keep = None

    def close(self):
        if self._h:
            self._lib.FLAC__stream_decoder_delete(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __getattr__(self, name):
        if name.startswith(("set_", "get_", "process_", "skip_", "finish", "flush", "reset")):
            prefix = "flacamd_stream_decoder_" if name in ("set_device", "set_batch_frames") else "FLAC__stream_decoder_"
            fn = getattr(self._lib, prefix + name)
            if name.startswith("set_") and name not in ("set_metadata_respond_all", "set_metadata_ignore_all"):
                return lambda value: bool(fn(self._h, int(value)))
            if name.startswith("get_"):
                return lambda: fn(self._h)
            return lambda: bool(fn(self._h))
        raise AttributeError(name)

    def get_resolved_state_string(self):
        return self._lib.FLAC__stream_decoder_get_resolved_state_string(self._h).decode()

    def set_metadata_respond_application(self, app_id):
        return bool(self._lib.FLAC__stream_decoder_set_metadata_respond_application(self._h, bytes(app_id)))

    def set_metadata_ignore_application(self, app_id):
        return bool(self._lib.FLAC__stream_decoder_set_metadata_ignore_application(self._h, bytes(app_id)))

    def get_decode_position(self):
        v = C.c_uint64(0)
        return int(v.value) if self._lib.FLAC__stream_decoder_get_decode_position(self._h, C.byref(v)) else None

    def seek_absolute(self, sample):
        return bool(self._lib.FLAC__stream_decoder_seek_absolute(self._h, sample))

    # callbacks (override write_callback to stop early etc.)
    def write_callback(self, header, pcm):
        self.frames.append((header, pcm))
        return 0

    def _callbacks(self):
        def w(_d, frame, bufs, _cd):
            h = frame.contents.header
            hdr = dict(blocksize=h.blocksize, sample_rate=h.sample_rate, channels=h.channels, channel_assignment=h.channel_assignment,
                       bits_per_sample=h.bits_per_sample, number_type=h.number_type, sample_number=h.number.sample_number)
            pcm = np.stack([np.ctypeslib.as_array(bufs[c], shape=(h.blocksize,)).copy() for c in range(h.channels)])
            self.events.append(("frame", int(h.number.sample_number), int(h.blocksize)))
            return int(self.write_callback(hdr, pcm) or 0)

        def m(_d, md, _cd):
            if md.contents.type != META_STREAMINFO:
                self.metadata.append(_block_to_dict(md.contents))
                return
            si = md.contents.data.stream_info
            self.streaminfo = dict(min_blocksize=si.min_blocksize, max_blocksize=si.max_blocksize, min_framesize=si.min_framesize,
                                   max_framesize=si.max_framesize, sample_rate=si.sample_rate, channels=si.channels,
                                   bits_per_sample=si.bits_per_sample, total_samples=si.total_samples, md5sum=bytes(si.md5sum))

        def e(_d, status, _cd):
            self.errors.append(status)
            self.events.append(("error", int(status)))

        return D_WRITE_CB(w), D_META_CB(m), D_ERROR_CB(e)

    def init(self):
        def r(_d, buf, pbytes, _cd):
            want = pbytes[0] if self.read_chunk is None else min(pbytes[0], self.read_chunk)
            chunk = self.data[self.pos:self.pos + want]
            C.memmove(buf, chunk, len(chunk))
            self.pos += len(chunk)
            pbytes[0] = len(chunk)
            return 1 if len(chunk) == 0 else 0

        def s(_d, off, _cd):
            self.pos = off
            return 0

        def t(_d, poff, _cd):
            poff[0] = self.pos
            return 0

        def ln(_d, plen, _cd):
            plen[0] = len(self.data)
            return 0

        def eof(_d, _cd):
            return 1 if self.pos >= len(self.data) else 0

        wcb, mcb, ecb = self._callbacks()
        if self.seekable:
            cbs = (D_READ_CB(r), D_SEEK_CB(s), D_TELL_CB(t), D_LENGTH_CB(ln), D_EOF_CB(eof), wcb, mcb, ecb)
        else:
            cbs = (D_READ_CB(r), D_SEEK_CB(), D_TELL_CB(), D_LENGTH_CB(), D_EOF_CB(), wcb, mcb, ecb)
        status = int(self._lib.FLAC__stream_decoder_init_stream(self._h, *cbs, None))
        if status == DEC_INIT_OK:
            self._keep = cbs
        return status

    def init_file(self, filename):
        cbs = self._callbacks()
        status = int(self._lib.FLAC__stream_decoder_init_file(self._h, filename.encode(), *cbs, None))
        if status == DEC_INIT_OK:
            self._keep = cbs
        return status

    def pcm(self):
        """all delivered frames as (samples, channels) int32"""
        if not self.frames:
            return np.zeros((0, 0), np.int32)
        return np.concatenate([p for _, p in self.frames], axis=1).T.copy()
