"""CPU tests of the host-side metadata code (flac_amd/csrc/metadata_blocks.c): block byte images
and legality checks against the reference library's own FLAC__add_metadata_block and
FLAC__format_* (oracle/_ref/libflacref.so, built from the reference's sources) on seeded random
blocks.  No GPU work."""
import ctypes as C
import random

import pytest

import oracle_bindings as ob
from flac_amd import engine, stream

VENDOR = b"reference libFLAC 1.2.1 20070917"
P = stream.SEEKPOINT_PLACEHOLDER

pytestmark = pytest.mark.skipif(not ob.have_ref(), reason="oracle/_ref not built")


def _libs():
    mine = engine.lib()
    ref = ob.ref_lib()
    mine.flacamd_metadata_block_image.argtypes = [C.POINTER(stream.StreamMetadata), C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    ref.FLAC__bitwriter_new.restype = C.c_void_p
    ref.FLAC__bitwriter_init.argtypes = [C.c_void_p]
    ref.FLAC__bitwriter_delete.argtypes = [C.c_void_p]
    ref.FLAC__add_metadata_block.argtypes = [C.POINTER(stream.StreamMetadata), C.c_void_p]
    ref.FLAC__bitwriter_get_buffer.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    for lib in (mine, ref):
        lib.FLAC__format_seektable_is_legal.argtypes = [C.POINTER(stream.SeekTable)]
        lib.FLAC__format_seektable_sort.argtypes = [C.POINTER(stream.SeekTable)]
        lib.FLAC__format_seektable_sort.restype = C.c_uint
        lib.FLAC__format_vorbiscomment_entry_is_legal.argtypes = [C.c_char_p, C.c_uint]
        lib.FLAC__format_cuesheet_is_legal.argtypes = [C.POINTER(stream.CueSheet), C.c_int, C.POINTER(C.c_char_p)]
        lib.FLAC__format_picture_is_legal.argtypes = [C.POINTER(stream.Picture), C.POINTER(C.c_char_p)]
    return mine, ref


def _image(mine, block):
    out, n = C.c_void_p(), C.c_size_t()
    assert mine.flacamd_metadata_block_image(C.byref(block), VENDOR, C.byref(out), C.byref(n)) == 0
    data = C.string_at(out, n.value)
    C.CDLL(None).free(out)
    return data


def _ref_image(ref, block):
    bw = C.c_void_p(ref.FLAC__bitwriter_new())
    assert ref.FLAC__bitwriter_init(bw)
    assert ref.FLAC__add_metadata_block(C.byref(block), bw)
    buf, n = C.c_void_p(), C.c_size_t()
    assert ref.FLAC__bitwriter_get_buffer(bw, C.byref(buf), C.byref(n))
    data = C.string_at(buf, n.value)
    ref.FLAC__bitwriter_delete(bw)
    return data


def _random_blocks(rng):
    yield stream.padding_block(rng.choice([0, 1, 17, 8192]))
    yield stream.application_block(bytes(rng.randrange(256) for _ in range(4)), bytes(rng.randrange(256) for _ in range(rng.randrange(0, 300))))
    yield stream.seektable_block([rng.choice([P, rng.randrange(1 << 40)]) for _ in range(rng.randrange(0, 20))])
    yield stream.vorbis_comment_block(["K%d=%s" % (i, "v" * rng.randrange(0, 50)) for i in range(rng.randrange(0, 6))])
    tracks = []
    for t in range(rng.randrange(1, 5)):
        idx = [(588 * rng.randrange(1000), j) for j in range(rng.randrange(0, 4))]
        tracks.append((588 * rng.randrange(10000), t + 1, bytes(rng.choice(b"ABC123") for _ in range(12)) if rng.random() < 0.5 else b"", rng.random() < 0.3, rng.random() < 0.3, idx))
    tracks.append((588 * 99999, 170, b"", False, False, []))
    yield stream.cuesheet_block(tracks, media_catalog_number=b"0123456789012" if rng.random() < 0.5 else b"", lead_in=88200, is_cd=rng.random() < 0.5)
    yield stream.picture_block(rng.randrange(0, 21), "image/png", "d" * rng.randrange(0, 30), rng.randrange(5000), rng.randrange(5000), 24, rng.randrange(3),
                               bytes(rng.randrange(256) for _ in range(rng.randrange(0, 500))))
    si = stream.StreamMetadata(type=stream.META_STREAMINFO, length=34)
    s = si.data.stream_info
    s.min_blocksize, s.max_blocksize, s.min_framesize, s.max_framesize = 4096, 4096, rng.randrange(1 << 24), rng.randrange(1 << 24)
    s.sample_rate, s.channels, s.bits_per_sample, s.total_samples = rng.randrange(1, 655350), rng.randrange(1, 9), rng.randrange(4, 33), rng.randrange(1 << 36)
    s.md5sum[:] = [rng.randrange(256) for _ in range(16)]
    yield si


def test_block_images_match_the_reference_writer():
    mine, ref = _libs()
    rng = random.Random(2024)
    for rnd in range(30):
        for b in _random_blocks(rng):
            b.is_last = rng.randrange(2)
            assert _image(mine, b) == _ref_image(ref, b), (rnd, b.type)


def test_vorbis_comment_vendor_string_is_replaced():
    mine, ref = _libs()
    b = stream.vorbis_comment_block(["A=b"])
    vend = b"someone else's encoder"
    buf, ptr = stream._bytes_ptr(vend)
    b.data.vorbis_comment.vendor_string.length, b.data.vorbis_comment.vendor_string.entry = len(vend), ptr
    b.length += len(vend)
    img = _image(mine, b)
    assert img == _ref_image(ref, b)
    assert int.from_bytes(img[1:4], "big") == 8 + len(VENDOR) + 4 + 3 and img[8:8 + len(VENDOR)] == VENDOR


def test_seektable_legality_and_sort_match_the_reference():
    mine, ref = _libs()
    rng = random.Random(7)
    for _ in range(300):
        vals = [rng.choice([P, rng.randrange(50), rng.randrange(1 << 45)]) for _ in range(rng.randrange(0, 25))]
        if rng.random() < 0.3:
            vals = sorted(set(v for v in vals if v != P)) + [P] * rng.randrange(3)
        a, b = stream.seektable_block(vals), stream.seektable_block(vals)
        for blk in (a, b):
            for i in range(len(vals)):
                blk.data.seek_table.points[i].stream_offset, blk.data.seek_table.points[i].frame_samples = 1000 + i, i
        ta, tb = a.data.seek_table, b.data.seek_table
        assert bool(mine.FLAC__format_seektable_is_legal(C.byref(ta))) == bool(ref.FLAC__format_seektable_is_legal(C.byref(tb)))
        assert mine.FLAC__format_seektable_sort(C.byref(ta)) == ref.FLAC__format_seektable_sort(C.byref(tb))
        pa = [(ta.points[i].sample_number, ta.points[i].stream_offset, ta.points[i].frame_samples) for i in range(len(vals))]
        pb = [(tb.points[i].sample_number, tb.points[i].stream_offset, tb.points[i].frame_samples) for i in range(len(vals))]
        # qsort is not stable: duplicates may keep either payload, so compare sample numbers everywhere and payloads where unique
        assert [p[0] for p in pa] == [p[0] for p in pb]
        for x, y in zip(pa, pb):
            if x[0] != P and vals.count(x[0]) == 1:
                assert x == y


def test_comment_picture_and_cuesheet_legality_match_the_reference():
    mine, ref = _libs()
    rng = random.Random(99)
    pool = [b"A", b"=", b" ", b"~", b"\x7f", b"\x1f", b"\xc3\xa9", b"\xc0\x80", b"\xe0\x80\x80", b"\xe2\x82\xac", b"\xed\xa0\x80", b"\xef\xbf\xbe",
            b"\xf0\x9f\x98\x80", b"\xf0\x80\x80\x80", b"\xf8\x88\x80\x80\x80", b"\xf8\x80\x80\x80\x80", b"\xfc\x84\x80\x80\x80\x80", b"\xfc\x80\x80\x80\x80\x80",
            b"\xff", b"\x80", b"\xc3", b"\xe2\x82"]
    for _ in range(2000):
        s = b"".join(rng.choice(pool) for _ in range(rng.randrange(0, 8)))
        assert bool(mine.FLAC__format_vorbiscomment_entry_is_legal(s + b"\0\0\0\0\0\0", len(s))) == bool(ref.FLAC__format_vorbiscomment_entry_is_legal(s + b"\0\0\0\0\0\0", len(s))), s
        desc = s.replace(b"\0", b"")
        pic = stream.picture_block(3, rng.choice(["image/png", "image/\x7f", "a\x1fb", ""]), desc, 1, 1, 1, 0, b"")
        assert bool(mine.FLAC__format_picture_is_legal(C.byref(pic.data.picture), None)) == bool(ref.FLAC__format_picture_is_legal(C.byref(pic.data.picture), None)), desc
    for _ in range(500):
        tracks = []
        for t in range(rng.randrange(0, 4)):
            idx = [(rng.choice([588, 1, 0]) * rng.randrange(100), rng.choice([j, j + 1, 0, 1])) for j in range(rng.randrange(0, 3))]
            tracks.append((rng.choice([588, 1]) * rng.randrange(1000), rng.choice([0, t + 1, 100, 170]), b"", False, False, idx))
        if rng.random() < 0.8:
            tracks.append((rng.choice([588, 7]) * 5000, rng.choice([170, 170, 5]), b"", False, False, []))
        if not tracks:
            continue
        blk = stream.cuesheet_block(tracks, lead_in=rng.choice([88200, 88199, 100, 588 * 200]))
        for cd in (0, 1):
            va, vb = C.c_char_p(), C.c_char_p()
            ra = mine.FLAC__format_cuesheet_is_legal(C.byref(blk.data.cue_sheet), cd, C.byref(va))
            rb = ref.FLAC__format_cuesheet_is_legal(C.byref(blk.data.cue_sheet), cd, C.byref(vb))
            assert bool(ra) == bool(rb) and va.value == vb.value, tracks
