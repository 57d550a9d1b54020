"""ctypes binding of the whole-file front end (include/flac_amd_files.h): encode a WAVE / AIFF /
raw PCM file to the .flac the reference's command-line tool would write, decode a .flac to
WAVE / AIFF / raw.  All work happens in libflacamd.so."""
import ctypes as C

from . import engine

AUTO, RAW, WAVE, AIFF, RF64, WAVE64, FLAC = range(7)


class FileOptions(C.Structure):
    _fields_ = [("compression_level", C.c_uint), ("verify", C.c_int), ("lax", C.c_int), ("padding", C.c_int), ("seekpoints", C.c_char_p),
                ("tags", C.POINTER(C.c_char_p)), ("num_tags", C.c_uint), ("blocksize", C.c_uint), ("input_format", C.c_int),
                ("raw_channels", C.c_uint), ("raw_bps", C.c_uint), ("raw_sample_rate", C.c_uint), ("raw_big_endian", C.c_int), ("raw_unsigned", C.c_int),
                ("device", C.c_int), ("batch_frames", C.c_uint), ("skip", C.c_char_p), ("until", C.c_char_p)]


class FileResult(C.Structure):
    _fields_ = [("samples", C.c_uint64), ("bytes_written", C.c_uint64), ("channels", C.c_uint), ("bits_per_sample", C.c_uint),
                ("sample_rate", C.c_uint), ("md5_ok", C.c_int), ("error", C.c_char * 256)]


class FileError(RuntimeError):
    def __init__(self, code, text):
        super().__init__("%s (%d)" % (text, code))
        self.code, self.text = code, text


_bound = False


def _lib():
    global _bound
    L = engine.lib()
    if not _bound:
        L.flacamd_file_options_init.argtypes = [C.POINTER(FileOptions)]
        L.flacamd_file_options_init.restype = None
        L.flacamd_encode_file.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(FileOptions), C.POINTER(FileResult)]
        L.flacamd_decode_file.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.POINTER(FileResult)]
        L.flacamd_decode_file_range.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_char_p, C.c_char_p, C.POINTER(FileResult)]
        _bound = True
    return L


def _result(rc, res):
    if rc != 0:
        raise FileError(rc, res.error.decode(errors="replace"))
    return dict(samples=res.samples, bytes_written=res.bytes_written, channels=res.channels, bits_per_sample=res.bits_per_sample,
                sample_rate=res.sample_rate, md5_ok=bool(res.md5_ok))


def encode_file(in_path, out_path, level=5, verify=False, lax=False, padding=-1, seekpoints=None, tags=(), blocksize=0,
                input_format=AUTO, raw=None, device=0, batch_frames=0, skip=None, until=None):
    """raw: dict(channels=, bps=, sample_rate=, big_endian=False, unsigned=False) with input_format=RAW.
    seekpoints: None for the tool's default, "" for none, else ';'-terminated -S values ("10s;", "X;", "4x;", "1234;")."""
    L = _lib()
    o = FileOptions()
    L.flacamd_file_options_init(C.byref(o))
    o.compression_level, o.verify, o.lax, o.padding, o.blocksize = level, int(verify), int(lax), padding, blocksize
    o.seekpoints = None if seekpoints is None else seekpoints.encode()
    enc = [t.encode() if isinstance(t, str) else bytes(t) for t in tags]
    arr = (C.c_char_p * max(1, len(enc)))(*enc)
    o.tags, o.num_tags = C.cast(arr, C.POINTER(C.c_char_p)), len(enc)
    o.input_format, o.device, o.batch_frames = input_format, device, batch_frames
    o.skip = None if skip is None else str(skip).encode()      # --skip / --until as the tool takes them
    o.until = None if until is None else str(until).encode()
    if raw:
        o.raw_channels, o.raw_bps, o.raw_sample_rate = raw["channels"], raw["bps"], raw["sample_rate"]
        o.raw_big_endian, o.raw_unsigned = int(raw.get("big_endian", False)), int(raw.get("unsigned", False))
    res = FileResult()
    return _result(L.flacamd_encode_file(str(in_path).encode(), str(out_path).encode(), C.byref(o), C.byref(res)), res)


def decode_file(in_path, out_path, output_format=WAVE, device=0, skip=None, until=None):
    L = _lib()
    res = FileResult()
    if skip is None and until is None:
        return _result(L.flacamd_decode_file(str(in_path).encode(), str(out_path).encode(), output_format, device, C.byref(res)), res)
    return _result(L.flacamd_decode_file_range(str(in_path).encode(), str(out_path).encode(), output_format, device,
                                               None if skip is None else str(skip).encode(), None if until is None else str(until).encode(), C.byref(res)), res)


def main(argv=None):
    """python -m flac_amd.files [-d] [-0..-8] [-V] [--lax] [-P n | --no-padding] [-S spec]... [-T tag]... [-b n] -o OUT IN"""
    import argparse
    ap = argparse.ArgumentParser(prog="flac_amd.files", description="encode WAVE/AIFF to FLAC, or decode (-d), on an MI355X")
    ap.add_argument("input")
    ap.add_argument("-o", "--output-name", required=True)
    ap.add_argument("-d", "--decode", action="store_true")
    for lv in range(9):
        ap.add_argument("-%d" % lv, dest="level", action="store_const", const=lv)
    ap.add_argument("-V", "--verify", action="store_true")
    ap.add_argument("--lax", action="store_true")
    ap.add_argument("-P", "--padding", type=int, default=-1)
    ap.add_argument("--no-padding", dest="padding", action="store_const", const=0)
    ap.add_argument("-S", "--seekpoint", action="append")
    ap.add_argument("--no-seektable", action="store_true")
    ap.add_argument("-T", "--tag", action="append", default=[])
    ap.add_argument("-b", "--blocksize", type=int, default=0)
    ap.add_argument("--force-aiff-format", action="store_true")
    ap.add_argument("--force-rf64-format", action="store_true")
    ap.add_argument("--force-wave64-format", action="store_true")
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--skip")
    ap.add_argument("--until")
    a = ap.parse_args(argv)
    try:
        if a.decode:
            low = a.output_name.lower()   # the tool's rules (src/flac/main.c:2020-2040)
            fmt = (AIFF if a.force_aiff_format or low.endswith((".aif", ".aiff")) else RF64 if a.force_rf64_format or low.endswith(".rf64")
                   else WAVE64 if a.force_wave64_format or low.endswith(".w64") else WAVE)
            r = decode_file(a.input, a.output_name, fmt, device=a.device, skip=a.skip, until=a.until)
        else:
            spec = "" if a.no_seektable else None if not a.seekpoint else "".join(s + ";" for s in a.seekpoint)
            r = encode_file(a.input, a.output_name, level=5 if a.level is None else a.level, verify=a.verify, lax=a.lax, padding=a.padding,
                            seekpoints=spec, tags=a.tag, blocksize=a.blocksize, device=a.device, skip=a.skip, until=a.until)
    except FileError as e:
        print("ERROR: %s" % e)
        return 1
    print("%s: %d samples, %d bytes" % (a.output_name, r["samples"], r["bytes_written"]))
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
