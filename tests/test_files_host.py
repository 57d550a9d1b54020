"""CPU tests of the whole-file front end's container parsing: everything that is rejected before
an encoder is created needs no GPU (flac_amd/csrc/file_frontend.c)."""
import struct

import pytest

from flac_amd import files


def _riff(chunks):
    body = b"WAVE" + b"".join(cid + struct.pack("<I", len(d)) + d + (b"\0" if len(d) & 1 else b"") for cid, d in chunks)
    return b"RIFF" + struct.pack("<I", len(body)) + body


FMT16 = struct.pack("<HHIIHH", 1, 2, 44100, 176400, 4, 16)
_GUID = b"\xF3\xAC\xD3\x11\xD1\x8C\x00\xC0\x4F\x8E\xDB\x8A"


def _rf64(chunks):
    body = b"WAVE" + b"".join(cid + struct.pack("<I", n if n is not None else len(d)) + d + (b"\0" if len(d) & 1 else b"") for cid, d, n in chunks)
    return b"RF64" + struct.pack("<I", 0xFFFFFFFF) + body


def _w64(chunks):
    body = b"wave" + _GUID + b"".join(cid + _GUID + struct.pack("<Q", n if n is not None else 24 + len(d)) + d + b"\0" * (-len(d) % 8) for cid, d, n in chunks)
    return b"riff\x2E\x91\xCF\x11\xD6\xA5\x28\xDB\x04\xC1\x00\x00" + struct.pack("<Q", 24 + len(body)) + body


DS64 = struct.pack("<QQQI", 100, 8, 2, 0)


@pytest.mark.parametrize("blob,text", [
    (b"OggS" + b"\0" * 40, "neither WAVE, RF64, Wave64 nor AIFF"),
    (_riff([(b"data", b"\0" * 8)]), "'data' chunk before 'fmt'"),
    (_riff([(b"fmt ", FMT16)]), "didn't find data chunk"),
    (_riff([(b"LIST", b"abcd")]), "didn't find fmt chunk"),
    (_riff([(b"fmt ", FMT16), (b"fmt ", FMT16)]), "multiple 'fmt ' chunks"),
    (_riff([(b"fmt ", FMT16[:14])]), "non-standard 'fmt ' chunk"),
    (_riff([(b"fmt ", struct.pack("<HHIIHH", 3, 2, 44100, 352800, 8, 32)), (b"data", b"\0" * 8)]), "unsupported format type 3"),
    (_riff([(b"fmt ", struct.pack("<HHIIHH", 1, 2, 44100, 176400, 4, 12)), (b"data", b"\0" * 8)]), "bits-per-sample=12"),
    (_riff([(b"fmt ", struct.pack("<HHIIHH", 1, 4, 44100, 352800, 8, 16)), (b"data", b"\0" * 8)]), ">2 channels"),
    (_riff([(b"fmt ", FMT16), (b"data", b"")]), "size of 0"),
    (_riff([(b"fmt ", struct.pack("<HHIIHH", 1, 2, 44100, 352800, 8, 32)), (b"data", b"\0" * 8)]), "unsupported bits-per-sample 32"),
    (_riff([(b"fmt ", struct.pack("<HHIIHHHHI", 65534, 2, 44100, 176400, 4, 16, 22, 16, 0x30) + b"\x01" + b"\0" * 15), (b"data", b"\0" * 8)]), "unsupported channel mask"),
    (_riff([(b"fmt ", struct.pack("<HHIIHHHHI", 65534, 2, 44100, 176400, 4, 16, 22, 16, 0x7) + b"\x01" + b"\0" * 15), (b"data", b"\0" * 8)]), "extra bits"),
    (_riff([(b"fmt ", struct.pack("<HHIIHHHHI", 65534, 2, 44100, 176400, 4, 16, 22, 17, 0x3) + b"\x01" + b"\0" * 15), (b"data", b"\0" * 8)]), "wValidBitsPerSample"),
    (_rf64([(b"fmt ", FMT16, None), (b"data", b"\0" * 8, 0xFFFFFFFF)]), "no 'ds64' chunk"),
    (_rf64([(b"ds64", DS64, None), (b"ds64", DS64, None)]), "multiple 'ds64' chunks"),
    (_rf64([(b"fmt ", FMT16, None), (b"ds64", DS64, None)]), "'ds64' chunk appears after"),
    (_rf64([(b"ds64", DS64[:20], None)]), "non-standard 'ds64' chunk"),
    (_rf64([(b"ds64", struct.pack("<QQQI", 100, 0, 0, 0), None), (b"fmt ", FMT16, None), (b"data", b"", 0xFFFFFFFF)]), "size of 0"),
    (_w64([(b"data", b"\0" * 8, None)]), "'data' chunk before 'fmt'"),
    (_w64([(b"fmt ", FMT16, 16)]), "freakishly small Wave64 'fmt ' chunk"),
    (_w64([(b"fmt ", FMT16, 1 << 33)]), "freakishly large Wave64 'fmt ' chunk"),
    (_w64([(b"fmt ", FMT16, None), (b"junk", b"", 8)]), "freakishly small Wave64 chunk"),
    (_w64([(b"fmt ", FMT16, None), (b"data", b"", 10)]), "freakishly small Wave64 'data' chunk"),
    (_w64([(b"fmt ", FMT16, None), (b"data", b"", None)]), "size of 0"),
    (b"fLaC" + b"\x84\0\0\x04abcd", "first metadata block is not STREAMINFO"),
    (b"fLaC" + b"\x80\0\0\x22" + b"\x10\0\x10\0" + b"\0" * 6 + b"\x0a\xc4\x42\xf0\0\0\0\0" + b"\0" * 16, "unknown total samples"),
    (b"fLaC" + b"\x80\0\0\x22" + b"\x10\0\x10\0" + b"\0" * 6 + b"\x00\x00\x02\xf0\0\0\0\x64" + b"\0" * 16, "unsupported sample rate 0"),
    # a chunk length near 2^64 must not wrap the position back into (or before) the file
    (_w64([(b"fmt ", FMT16, None), (b"junk", b"", 0xFFFFFFFFFFFFFFD0), (b"data", b"\0" * 64, None)]), "didn't find data chunk"),
    (_w64([(b"junk", b"", 0xFFFFFFFFFFFFFFF8), (b"fmt ", FMT16, None), (b"data", b"\0" * 64, None)]), "didn't find fmt chunk"),
    (_w64([(b"fmt ", FMT16, None), (b"junk", b"", 1 << 63)]), "didn't find data chunk"),
    (_riff([(b"fmt ", FMT16), (b"LIST", b"abcd")])[:-2] , "didn't find data chunk"),
    (b"FORM" + struct.pack(">I", 4) + b"AIFF", "didn't find COMM"),
    (b"FORM" + struct.pack(">I", 30) + b"AIFF" + b"COMM" + struct.pack(">IHIH", 18, 4, 10, 16) + b"\x40\x0e\xac\x44" + b"\0" * 6, "unsupported number of channels 4"),
])
def test_inputs_the_tool_rejects(tmp_path, blob, text):
    src = tmp_path / "in.bin"
    src.write_bytes(blob)
    with pytest.raises(files.FileError) as ei:
        files.encode_file(src, tmp_path / "out.flac")
    assert text in ei.value.text and ei.value.code in (-1, -2, -8)


def test_missing_files_and_bad_arguments(tmp_path):
    with pytest.raises(files.FileError) as ei:
        files.encode_file(tmp_path / "nope.wav", tmp_path / "o.flac")
    assert "can't open input" in ei.value.text
    (tmp_path / "x.raw").write_bytes(b"\0" * 64)
    with pytest.raises(files.FileError) as ei:
        files.encode_file(tmp_path / "x.raw", tmp_path / "o.flac", input_format=files.RAW)
    assert "raw input needs" in ei.value.text
    with pytest.raises(files.FileError) as ei:
        files.decode_file(tmp_path / "x.flac", tmp_path / "o.wav", output_format=9)
    assert "unknown output format" in ei.value.text
