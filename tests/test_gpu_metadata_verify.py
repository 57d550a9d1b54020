"""GPU tests of client metadata blocks and verify mode of the libFLAC-style encoder object:
whole files against the reference's flac tool run with the options that make it build the
same blocks (src/flac/encode.c:2016-2053), and the verify failure paths
(stream_encoder.c:2396-2411, :4197-4250)."""
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

import oracle_bindings as ob
from flac_amd import stream, synth

pytestmark = pytest.mark.gpu

RAW = ["--silent", "--force-raw-format", "--endian=little", "--sign=signed", "--channels=2", "--bps=16", "--sample-rate=44100"]


def _tool(tmp_path, pcm, extra):
    if not os.path.exists(ob.REF_FLAC):
        pytest.skip("oracle/_ref/flac not built")
    raw = tmp_path / "in.raw"
    pcm.astype("<i2").tofile(raw)
    want = tmp_path / "ref.flac"
    subprocess.run([ob.REF_FLAC] + RAW + list(extra) + ["-o", str(want), str(raw)], check=True, timeout=120)
    return want.read_bytes()


def _encode_file(tmp_path, pcm, blocks, level=8, batch=5, verify=False, chunk=7000):
    e = stream.FileEncoder()
    assert e.set_compression_level(level) and e.set_total_samples_estimate(len(pcm)) and e.set_batch_frames(batch)
    assert e.set_verify(verify) and e.get_verify() == int(verify)
    assert e.set_metadata(blocks)
    out = tmp_path / "gpu.flac"
    assert e.init(str(out)) == stream.INIT_OK
    for i in range(0, len(pcm), chunk):
        assert e.process_interleaved(pcm[i:i + chunk])
    assert e.finish()
    e.close()
    return out.read_bytes()


@pytest.mark.parametrize("seconds,level", [(31.7, 8), (3.0, 5), (0.05, 0)])
def test_default_tool_layout_seektable_tags_padding(gpu_engine, tmp_path, seconds, level):
    """flac's defaults: a seek point every 10 s, the (empty) tag block, 8192 bytes of padding."""
    n = int(44100 * seconds)
    pcm = synth.music_like(seconds + 0.1, seed=11)[:n]
    want = _tool(tmp_path, pcm, ["-%d" % level])
    blocks = [stream.seektable_block(stream.spaced_seekpoints(441000, n)), stream.vorbis_comment_block([]), stream.padding_block(8192)]
    got = _encode_file(tmp_path, pcm, blocks, level=level)
    assert got == want


def test_tags_explicit_points_placeholders_and_small_padding(gpu_engine, tmp_path):
    n = 44100 * 6
    pcm = synth.music_like(6.1, seed=12)[:n]
    tags = ["ARTIST=Somebody", "TITLE=Grüße – 音楽", "EMPTY="]
    opts = ["-8", "--no-utf8-convert", "--padding=17", "-S", "100000", "-S", "5000", "-S", "X", "-S", "5001", "-S", "X", "-S", "263000"]
    for t in tags:
        opts += ["-T", t]
    want = _tool(tmp_path, pcm, opts)
    P = stream.SEEKPOINT_PLACEHOLDER
    # the tool sorts and compacts its template (grabbag/seektable.c:128); 5000 and 5001 land in one frame and are merged at finish
    blocks = [stream.seektable_block([5000, 5001, 100000, 263000, P, P]), stream.vorbis_comment_block(tags), stream.padding_block(17)]
    got = _encode_file(tmp_path, pcm, blocks, batch=3)
    assert got == want


def _png(width, height):
    def chunk(kind, data):
        return struct.pack(">I", len(data)) + kind + data + struct.pack(">I", zlib.crc32(kind + data))
    raw = b"".join(b"\0" + bytes([i & 255, 0, 255 - (i & 255)] * width) for i in range(height))
    return b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", width, height, 8, 2, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(raw)) + chunk(b"IEND", b"")


def test_pictures_and_application_order(gpu_engine, tmp_path):
    n = 44100 * 2
    pcm = synth.music_like(2.1, seed=13)[:n]
    png = _png(32, 32)
    f = tmp_path / "icon.png"
    f.write_bytes(png)
    want = _tool(tmp_path, pcm, ["-5", "--no-seektable", "--padding=4", "--picture=1||the icon||%s" % f, "--picture=3|image/png|front|32x32x24|%s" % f])
    blocks = [stream.vorbis_comment_block([]), stream.picture_block(1, "image/png", "the icon", 32, 32, 24, 0, png),
              stream.picture_block(3, "image/png", "front", 32, 32, 24, 0, png), stream.padding_block(4)]
    got = _encode_file(tmp_path, pcm, blocks, level=5)
    assert got == want
    # an APPLICATION block and an unknown block type travel through unchanged, in the order given
    app = stream.application_block(b"tEsT", b"payload bytes")
    unk = stream.StreamMetadata(type=99, length=3)
    buf, ptr = stream._bytes_ptr(b"xyz")
    unk.data.unknown.data = ptr
    got = _encode_file(tmp_path, pcm, [app, unk], level=5)
    head = got[:4 + 38]
    rest = got[42:]
    assert head[:4] == b"fLaC" and rest[0] == 4 and rest[4 + 40:4 + 40 + 21] == bytes([2, 0, 0, 17]) + b"tEsT" + b"payload bytes"
    assert rest[4 + 40 + 21:4 + 40 + 21 + 7] == bytes([0x80 | 99, 0, 0, 3]) + b"xyz"


def test_cuesheet_block_and_cued_seekpoints(gpu_engine, tmp_path):
    n = 588 * 450          # 6 s, CD frames
    pcm = synth.music_like(6.1, seed=14)[:n]
    cue = tmp_path / "disc.cue"
    cue.write_text('CATALOG 1234567890123\nFILE "in.wav" WAVE\n  TRACK 01 AUDIO\n    ISRC ABCDE1234567\n    INDEX 01 00:00:00\n'
                   '  TRACK 02 AUDIO\n    FLAGS PRE\n    INDEX 00 00:02:00\n    INDEX 01 00:03:37\n')
    want = _tool(tmp_path, pcm, ["-8", "--no-padding", "--cuesheet=%s" % cue, "-S", "1x"])
    t2, i1 = 588 * 150, 588 * (75 + 37)
    tracks = [(0, 1, b"ABCDE1234567", False, False, [(0, 1)]), (t2, 2, b"", False, True, [(0, 0), (i1, 1)]), (n, 170, b"", False, False, [])]
    blocks = [stream.seektable_block([0, t2, t2 + i1]), stream.cuesheet_block(tracks, media_catalog_number=b"1234567890123"), stream.vorbis_comment_block([])]
    got = _encode_file(tmp_path, pcm, blocks)
    assert got == want


def test_invalid_metadata_is_refused_at_init(gpu_engine):
    def status(blocks):
        e = stream.MemoryEncoder()
        assert e.set_metadata(blocks)
        st = e.init()
        e.close()
        return st
    bad_table = stream.seektable_block([100, 100])
    assert status([stream.StreamMetadata(type=stream.META_STREAMINFO, length=34)]) == stream.INIT_INVALID_METADATA
    assert status([bad_table]) == stream.INIT_INVALID_METADATA
    assert status([stream.seektable_block([1]), stream.seektable_block([2])]) == stream.INIT_INVALID_METADATA
    assert status([stream.vorbis_comment_block([]), stream.vorbis_comment_block([])]) == stream.INIT_INVALID_METADATA
    assert status([stream.picture_block(1, "image/jpeg", "", 32, 32, 24, 0, b"x")]) == stream.INIT_INVALID_METADATA
    assert status([stream.picture_block(2, "image/png", "", 1, 1, 24, 0, b"x")] * 2) == stream.INIT_INVALID_METADATA
    assert status([stream.picture_block(3, "image/png", b"\xff\xfe", 1, 1, 24, 0, b"x")]) == stream.INIT_INVALID_METADATA
    assert status([stream.cuesheet_block([(0, 0, b"", False, False, [])], is_cd=False)]) == stream.INIT_INVALID_METADATA
    assert status([stream.padding_block(10), stream.padding_block(0)]) == stream.INIT_OK


def test_verify_passes_and_output_is_unchanged(gpu_engine, tmp_path):
    n = 4096 * 9 + 100
    pcm = synth.music_like(1.0, seed=15)[:n]
    plain = _encode_file(tmp_path, pcm, [], verify=False, batch=4)
    checked = _encode_file(tmp_path, pcm, [], verify=True, batch=4)
    assert plain == checked
    e = stream.MemoryEncoder()
    assert e.set_verify(True) and e.set_channels(8) and e.set_bits_per_sample(24) and e.set_sample_rate(96000) and e.set_compression_level(8)
    assert e.init() == stream.INIT_OK
    assert e.get_verify_decoder_state() == stream.DEC_SEARCH_FOR_FRAME_SYNC
    assert e.process_interleaved(synth.multichannel_24(0.2, channels=8, seed=3)) and e.finish()
    assert e.get_verify_decoder_state() == stream.DEC_UNINITIALIZED
    e.close()


@pytest.mark.parametrize("where,channel", [(4096 * 5 + 17, 1), (0, 0), (4096 * 9 + 99, 1)])
def test_verify_reports_an_audio_mismatch(gpu_engine, where, channel):
    n = 4096 * 9 + 100
    pcm = synth.music_like(1.0, seed=16)[:n].astype(np.int32)
    e = stream.MemoryEncoder()
    assert e.set_verify(True) and e.set_compression_level(5) and e.set_batch_frames(4)
    assert e.inject_verify_fault(where, channel, 0x40)
    assert e.init() == stream.INIT_OK
    ok = e.process_interleaved(pcm)
    ok = ok and e.finish()
    assert not ok
    assert e.get_state() == stream.VERIFY_MISMATCH
    assert e.get_resolved_state_string() == "FLAC__STREAM_ENCODER_VERIFY_MISMATCH_IN_AUDIO_DATA"
    bad_frame = where // 4096
    bs = min(4096, n - bad_frame * 4096)
    st = e.get_verify_decoder_error_stats()
    assert st == dict(absolute_sample=where, frame_number=(bad_frame * 4096) // bs, channel=channel, sample=where % 4096,
                      expected=int(pcm[where, channel]) ^ 0x40, got=int(pcm[where, channel]))
    # every frame ahead of the bad one reached the client, the bad one and its successors did not
    assert [w[2] for w in e.writes if w[1] > 0] == list(range(bad_frame))
    assert not e.finish() or e.get_state() != stream.OK
    e.close()


def test_verify_reports_a_frame_that_does_not_decode(gpu_engine):
    n = 4096 * 7 + 1
    pcm = synth.music_like(1.0, seed=17)[:n]
    e = stream.MemoryEncoder()
    assert e.set_verify(True) and e.set_compression_level(8) and e.set_batch_frames(7)
    assert e.inject_frame_fault(3, 40, 0x01)
    assert e.init() == stream.INIT_OK
    assert not e.process_interleaved(pcm)      # one full batch plus the sample that releases it
    assert e.get_state() == stream.VERIFY_DECODER_ERROR
    assert e.get_verify_decoder_state() == stream.DEC_SEARCH_FOR_FRAME_SYNC
    assert e.get_resolved_state_string() == "FLAC__STREAM_DECODER_SEARCH_FOR_FRAME_SYNC"   # the decoder's state is reported (:1804-1807)
    assert [w[2] for w in e.writes if w[1] > 0] == [0, 1, 2]
    assert not e.finish() and e.get_state() == stream.VERIFY_DECODER_ERROR
    assert e.get_resolved_state_string() == "FLAC__STREAM_DECODER_UNINITIALIZED"
    e.close()


def test_decoder_delivers_the_blocks_the_filter_selects(gpu_engine, tmp_path):
    """Round trip of every block type through the decoder's metadata filter (stream_decoder.c:762-877, :1427-1602)."""
    n = 588 * 100
    pcm = synth.music_like(1.5, seed=18)[:n]
    png = _png(4, 4)
    tracks = [(0, 1, b"ABCDE1234567", False, True, [(0, 1), (588, 2)]), (n, 170, b"", False, False, [])]
    blocks = [stream.seektable_block([0, 30000, stream.SEEKPOINT_PLACEHOLDER]), stream.application_block(b"aaaa", b"first"),
              stream.vorbis_comment_block(["TITLE=x", "EMPTY=", "K=ü"]), stream.cuesheet_block(tracks, media_catalog_number=b"42"),
              stream.application_block(b"bbbb", b""), stream.picture_block(3, "image/png", "cover ü", 4, 4, 24, 0, png), stream.padding_block(33)]
    data = _encode_file(tmp_path, pcm, blocks, level=5)

    def run(setup):
        d = stream.StreamDecoder(data)
        setup(d)
        assert d.init() == stream.DEC_INIT_OK and d.process_until_end_of_stream() and d.finish()
        out = d.metadata, d.streaminfo, d.pcm()
        d.close()
        return out

    meta, si, got = run(lambda d: d.set_metadata_respond_all())
    assert np.array_equal(got, pcm) and si["total_samples"] == n
    assert [m["type"] for m in meta] == [3, 2, 4, 5, 2, 6, 1] and [m["is_last"] for m in meta] == [False] * 6 + [True]
    assert meta[0]["points"] == [(0, 0, 4096), (28672, meta[0]["points"][1][1], 4096), (stream.SEEKPOINT_PLACEHOLDER, 0, 0)]
    assert (meta[1]["id"], meta[1]["data"], meta[4]["id"], meta[4]["data"]) == (b"aaaa", b"first", b"bbbb", b"")
    assert meta[2]["vendor"] == b"reference libFLAC 1.2.1 20070917" and meta[2]["comments"] == [b"TITLE=x", b"EMPTY=", "K=ü".encode()]
    cs = meta[3]
    assert (cs["media_catalog_number"], cs["lead_in"], cs["is_cd"]) == (b"42", 88200, True) and cs["tracks"] == tracks
    pic = meta[5]
    assert (pic["picture_type"], pic["mime_type"], pic["description"], pic["width"], pic["height"], pic["depth"], pic["colors"], pic["data"]) == \
        (3, b"image/png", "cover ü".encode(), 4, 4, 24, 0, png)
    assert meta[6]["length"] == 33
    # default filter: STREAMINFO only; ignore_all: nothing at all
    meta, si, _ = run(lambda d: None)
    assert meta == [] and si is not None
    meta, si, _ = run(lambda d: d.set_metadata_ignore_all())
    assert meta == [] and si is None
    # application ids invert the APPLICATION setting (:780-865)
    meta, _, _ = run(lambda d: d.set_metadata_respond_application(b"bbbb"))
    assert [(m["type"], m["id"]) for m in meta] == [(2, b"bbbb")]
    meta, _, _ = run(lambda d: (d.set_metadata_respond(stream.META_APPLICATION), d.set_metadata_respond(stream.META_VORBIS_COMMENT), d.set_metadata_ignore_application(b"bbbb")))
    assert [m["type"] for m in meta] == [2, 4] and meta[0]["id"] == b"aaaa"
