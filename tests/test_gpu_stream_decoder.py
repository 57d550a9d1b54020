"""GPU tests of the libFLAC-style stream decoder object (include/flac_amd_stream.h): whole
streams as the reference writes them go in through the read callback; frames, headers,
STREAMINFO, MD5 verdict and seek behaviour come out as with the reference decoder."""
import os

import numpy as np
import pytest

import oracle_bindings as ob
from golden_util import load_golden
from flac_amd import stream, synth
from test_gpu_stream import VC_BLOCK

pytestmark = pytest.mark.gpu


def _stream_of(g, md5=None):
    m = g["meta"]
    sizes = [len(f) for f in g["frames"]]
    first = _blocksize_of(g)
    head = ob.stream_head(first, first, min(sizes), max(sizes), m["rate"], m["channels"], m["bps"], len(g["pcm"]), md5 or g["md5"], is_last=False)
    return head + VC_BLOCK + b"".join(g["frames"])


def _blocksize_of(g):
    # blocksize of the stream = samples of the first frame unless the stream is a single short frame
    m = g["meta"]
    p = m["params"]
    if p.get("blocksize"):
        return p["blocksize"]
    lvl = p.get("level", 8)
    lpc = p.get("max_lpc_order", {0: 0, 1: 0, 2: 0}.get(lvl, 8))
    return 4096 if lpc else 1152


@pytest.mark.parametrize("name,batch,chunk", [
    ("st16-level5", 2048, None), ("st16-level8", 2, 1000), ("st16-8-bs17", 64, 333), ("mono16-8", 1, None), ("8ch24-8", 3, 5000),
    ("sine24-st", 2048, 4096), ("loose-4", 5, None), ("len5", 4, 1), ("len4097", 1, 7), ("wasted", 2048, None), ("white", 2, None),
    ("silence", 2048, 16), ("st16-level0", 7, None),
])
def test_decoder_object_returns_the_original_pcm(gpu_engine, name, batch, chunk):
    g = load_golden(name)
    m = g["meta"]
    data = _stream_of(g)
    d = stream.StreamDecoder(data, seekable=False, read_chunk=chunk)
    assert d.set_md5_checking(1) and d.set_batch_frames(batch)
    assert d.init() == stream.DEC_INIT_OK and d.get_state() == stream.DEC_SEARCH_FOR_METADATA
    assert d.process_until_end_of_metadata() and d.get_state() == stream.DEC_SEARCH_FOR_FRAME_SYNC
    si = d.streaminfo
    assert (si["channels"], si["bits_per_sample"], si["sample_rate"], si["total_samples"], si["md5sum"]) == (m["channels"], m["bps"], m["rate"], len(g["pcm"]), g["md5"])
    assert d.process_until_end_of_stream() and d.get_state() == stream.DEC_END_OF_STREAM
    assert d.errors == []
    assert np.array_equal(d.pcm(), g["pcm"])
    # per-frame header fields as write_audio_frame_to_client_ hands them over
    bs = si["max_blocksize"]
    pos = 0
    for hdr, pcm in d.frames:
        assert hdr["sample_number"] == pos and hdr["number_type"] == 1 and hdr["channels"] == m["channels"] and hdr["bits_per_sample"] == m["bps"]
        assert hdr["blocksize"] == pcm.shape[1] == min(bs, len(g["pcm"]) - pos) and hdr["sample_rate"] == m["rate"]
        pos += hdr["blocksize"]
    assert len(d.frames) == len(g["frames"])
    assert d.finish() and d.get_state() == stream.DEC_UNINITIALIZED      # MD5 matches
    d.close()


def test_md5_mismatch_is_reported_by_finish(gpu_engine):
    g = load_golden("st16-level3")
    d = stream.StreamDecoder(_stream_of(g, md5=bytes(range(1, 17))), seekable=False)
    assert d.set_md5_checking(1) and d.init() == 0
    assert d.process_until_end_of_stream()
    assert np.array_equal(d.pcm(), g["pcm"])
    assert not d.finish()                                               # stream_decoder.c:685-693
    d.close()


def test_process_single_steps_and_skip(gpu_engine):
    g = load_golden("st16-level5")
    d = stream.StreamDecoder(_stream_of(g), seekable=True)
    assert d.set_batch_frames(2) and d.init() == 0
    assert d.process_single() and d.get_state() == stream.DEC_READ_METADATA            # found "fLaC"
    assert d.process_single() and d.streaminfo is not None and d.get_state() == stream.DEC_READ_METADATA   # STREAMINFO
    assert d.process_single() and d.get_state() == stream.DEC_SEARCH_FOR_FRAME_SYNC    # VORBIS_COMMENT (last)
    assert d.process_single() and len(d.frames) == 1
    assert d.skip_single_frame() and len(d.frames) == 1
    assert d.process_single() and len(d.frames) == 2 and d.frames[1][0]["sample_number"] == 2 * 4096
    assert d.get_channels() == 2 and d.get_bits_per_sample() == 16 and d.get_sample_rate() == 44100 and d.get_blocksize() == 4096
    sizes = [len(f) for f in g["frames"]]
    assert d.get_decode_position() == 42 + len(VC_BLOCK) + sum(sizes[:3])
    assert d.process_until_end_of_stream() and d.get_state() == stream.DEC_END_OF_STREAM
    assert d.process_single()                                            # stays true at end of stream (:1019-1021)
    assert d.finish()
    d.close()


@pytest.mark.parametrize("target", [0, 1, 4095, 4096, 5000, 12287, 12388 - 1])
def test_seek_absolute_lands_on_the_sample(gpu_engine, target):
    g = load_golden("st16-level8")
    n = len(g["pcm"])
    assert target < n
    d = stream.StreamDecoder(_stream_of(g), seekable=True)
    assert d.set_md5_checking(1) and d.set_batch_frames(2) and d.init() == 0
    assert d.process_until_end_of_metadata() and d.process_single()
    d.frames = []
    assert d.seek_absolute(target)
    # the frame that holds the target arrives trimmed to start exactly there (stream_decoder.c:2920-2956)
    assert len(d.frames) == 1 and d.frames[0][0]["sample_number"] == target
    assert d.process_until_end_of_stream()
    assert np.array_equal(d.pcm(), g["pcm"][target:])
    assert not d.seek_absolute(n)                                        # past the end: refused
    assert d.finish()                                                    # MD5 checking was switched off by the seek
    d.close()


def test_id3v2_tag_and_trailing_junk(gpu_engine):
    g = load_golden("mono16-8")
    body = _stream_of(g)
    tag = b"ID3\x03\x00\x00" + bytes([0, 0, 1, 5]) + bytes(133)           # synchsafe size 133
    d = stream.StreamDecoder(tag + body + b"TAG" + bytes(125), seekable=False)
    assert d.set_md5_checking(1) and d.init() == 0
    assert d.process_until_end_of_stream() and d.get_state() == stream.DEC_END_OF_STREAM
    assert np.array_equal(d.pcm(), g["pcm"])
    assert d.finish()
    d.close()


def test_damaged_frame_is_reported_not_hidden(gpu_engine):
    # as the reference (stream_decoder.c:2111-2119): an error report, the frame delivered as zeros, decoding goes on
    g = load_golden("st16-level5")
    data = bytearray(_stream_of(g))
    data[42 + len(VC_BLOCK) + len(g["frames"][0]) + 40] ^= 0x10           # inside frame 1
    d = stream.StreamDecoder(bytes(data), seekable=False)
    assert d.set_md5_checking(True)
    assert d.init() == 0
    assert d.process_until_end_of_stream()
    assert d.errors == [stream.ERR_FRAME_CRC_MISMATCH] and d.get_state() == stream.DEC_END_OF_STREAM
    assert len(d.frames) == len(g["frames"]) and not d.frames[1][1].any() and d.frames[0][1].any() and d.frames[2][1].any()
    assert not d.finish()                                                  # the zeros went through the MD5
    d.close()


def test_write_abort_and_reinit(gpu_engine):
    class Stop(stream.StreamDecoder):
        def write_callback(self, header, pcm):
            super().write_callback(header, pcm)
            return 1 if len(self.frames) == 2 else 0

    g = load_golden("st16-level3")
    d = Stop(_stream_of(g), seekable=True)
    assert d.init() == 0 and d.init() == stream.DEC_INIT_ALREADY_INITIALIZED
    assert not d.process_until_end_of_stream() and d.get_state() == stream.DEC_READ_FRAME and len(d.frames) == 2   # stream_decoder.c:2138-2141
    assert d.finish()
    d.close()


def test_file_decoder_round_trip_with_reference_encoder(gpu_engine, tmp_path):
    """A file written by the reference library, read back through init_file."""
    if not ob.have_ref():
        pytest.skip("oracle/_ref not built")
    pcm = synth.music_like(2.0, seed=21)
    _, rp = ob.make_params(2, 16, 44100, level=6)
    ref = ob.ref_encode(rp, pcm)
    head = ob.stream_head(4096, 4096, ref["min_framesize"], ref["max_framesize"], 44100, 2, 16, len(pcm), ref["md5"], is_last=False)
    path = tmp_path / "ref.flac"
    path.write_bytes(head + VC_BLOCK + b"".join(ref["frames"]))
    d = stream.StreamDecoder()
    assert d.set_md5_checking(1) and d.init_file(str(path)) == 0
    assert d.process_until_end_of_stream()
    assert np.array_equal(d.pcm(), pcm)
    assert d.finish()
    d.close()


# ---- seeking by seek table and interpolation, bounded input, robustness against what the file claims -------------

class _CountingBytes:
    """the input of a decoder that counts the bytes asked of it"""

    def __init__(self, data):
        self.data, self.count = data, 0

    def __len__(self):
        return len(self.data)

    def __getitem__(self, sl):
        out = self.data[sl]
        self.count += len(out)
        return out


def _long_file(tmp_path, seconds, extra):
    import struct
    import subprocess
    ref = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "flac")
    if not os.path.exists(ref):
        pytest.skip("oracle/_ref/flac not built")
    pcm = synth.music_like(seconds, seed=321)
    data = pcm.astype("<i2").tobytes()
    wav = str(tmp_path / "long.wav")
    with open(wav, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", 36 + len(data)) + b"WAVEfmt " + struct.pack("<IHHIIHH", 16, 1, 2, 44100, 176400, 4, 16) + b"data" + struct.pack("<I", len(data)))
        f.write(data)
    out = str(tmp_path / "long.flac")
    subprocess.run([ref, "--silent", "-f", "-5"] + extra + ["-o", out, wav], check=True, timeout=300)
    return pcm, open(out, "rb").read()


@pytest.mark.parametrize("extra", [[], ["--no-seektable"], ["-S", "1s"]])
def test_seek_in_a_long_stream_reads_a_fraction_of_it(gpu_engine, tmp_path, extra):
    """seek_to_absolute_sample_ (stream_decoder.c:2979-3192) narrows the search with the seek table and by interpolation; the
    frames delivered after a seek are the ones the reference delivers, and the input read for it stays far below the stream's size."""
    pcm, data = _long_file(tmp_path, 240.0, extra)
    n = len(pcm)
    rng = np.random.default_rng(5)
    src = _CountingBytes(data)
    d = stream.StreamDecoder(src, seekable=True)
    assert d.set_batch_frames(64) and d.init() == 0
    assert d.process_until_end_of_metadata()
    for target in [n - 1, 0, n // 2, 4095, 4096] + [int(v) for v in rng.integers(0, n, 12)]:
        d.frames = []
        before = src.count
        assert d.seek_absolute(target), target
        assert len(d.frames) == 1 and d.frames[0][0]["sample_number"] == target
        got = d.frames[0][1].T
        assert np.array_equal(got, pcm[target:target + got.shape[0]])
        assert d.process_single() or True
        assert src.count - before < len(data) // 4, (target, src.count - before, len(data))
        if len(d.frames) > 1:
            h, p = d.frames[1]
            assert h["sample_number"] == target + got.shape[0] and np.array_equal(p.T, pcm[h["sample_number"]:h["sample_number"] + p.shape[1]])
    d.close()


def test_a_streaminfo_that_claims_huge_frames_still_decodes(gpu_engine):
    """max_framesize comes from the file and is advisory (the reference never sizes anything by it)."""
    g = load_golden("st16-level5")
    data = bytearray(_stream_of(g))
    data[4 + 4 + 7:4 + 4 + 10] = b"\xff\xff\xff"        # STREAMINFO max_framesize = 0xFFFFFF
    d = stream.StreamDecoder(bytes(data), seekable=False)
    assert d.set_md5_checking(1) and d.init() == 0
    assert d.process_until_end_of_stream() and d.get_state() == stream.DEC_END_OF_STREAM
    assert np.array_equal(d.pcm(), g["pcm"]) and d.finish()
    d.close()


def test_input_that_trickles_in_is_decoded_as_it_arrives(gpu_engine):
    """A source that hands over a few hundred bytes per read (a pipe, a live stream): frames come out as soon as they are
    complete, not when a whole window of megabytes has arrived."""
    if not ob.have_ref():
        pytest.skip("oracle/_ref not built")
    pcm = synth.music_like(3.0, seed=77)
    _, rp = ob.make_params(2, 16, 44100, level=8)
    r = ob.ref_encode(rp, pcm)
    data = ob.stream_head(4096, 4096, r["min_framesize"], r["max_framesize"], 44100, 2, 16, len(pcm), r["md5"], is_last=False) + r["stream"][42:]
    g = dict(pcm=pcm)

    class Slow(stream.StreamDecoder):
        def write_callback(self, header, pcm):
            self.seen.append(self.pos)           # how far the input had been read when this frame came out
            return super().write_callback(header, pcm)

    d = Slow(data, seekable=False, read_chunk=700)
    d.seen = []
    assert d.set_md5_checking(1) and d.init() == 0
    assert d.process_until_end_of_stream()
    assert np.array_equal(d.pcm(), g["pcm"]) and d.finish()
    assert len(data) > 300000 and d.seen[0] < 40000, "the first frame waited for a whole window"
    assert all(b - a < 60000 for a, b in zip(d.seen, d.seen[1:]))
    d.close()


def test_a_metadata_block_with_inconsistent_contents_is_passed_over(gpu_engine):
    g = load_golden("mono16-8")
    data = _stream_of(g)
    # a VORBIS_COMMENT whose vendor length runs past the block, between STREAMINFO and the frames
    bad = bytes([4, 0, 0, 12]) + (500).to_bytes(4, "little") + bytes(8)
    spliced = data[:42] + bad + data[42:]       # STREAMINFO, the bad block, then the stream's own (last) VORBIS_COMMENT and the frames
    d = stream.StreamDecoder(spliced, seekable=False)
    assert d.set_metadata_respond_all() and d.set_md5_checking(1) and d.init() == 0
    assert d.process_until_end_of_stream() and d.get_state() == stream.DEC_END_OF_STREAM
    assert np.array_equal(d.pcm(), g["pcm"]) and d.finish()
    assert d.errors == [1] and len(d.metadata) == 1      # BAD_HEADER reported, nothing handed out for the bad block; the good one arrives
    d.close()
