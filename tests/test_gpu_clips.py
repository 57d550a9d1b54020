"""Batch of short clips (mixed 16/24-bit, mono/stereo, 44.1/48 kHz) at -5: every clip must come
out as the complete stream the reference writes for it, and decode back bit-exactly."""
import numpy as np
import pytest

import oracle_bindings as ob
from flac_amd import clips, stream, synth
from test_gpu_stream import VC_BLOCK

pytestmark = pytest.mark.gpu


def _make_clips(count, seed):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(count):
        dur = float(rng.uniform(0.05, 1.2))
        bps = 16 if rng.random() < 0.5 else 24
        rate = 44100 if rng.random() < 0.5 else 48000
        ch = 1 if rng.random() < 0.5 else 2
        if bps == 16:
            pcm = synth.music_like(dur, rate=rate, seed=1000 + i)[:, :ch]
        else:
            pcm = synth.multichannel_24(dur, rate=rate, channels=ch, seed=2000 + i)
        out.append((np.ascontiguousarray(pcm), bps, rate))
    return out


def test_clip_batch_streams_match_reference(gpu_engine):
    batch = _make_clips(20, seed=777)
    got = clips.encode_clips(batch, level=5)
    assert len(got) == len(batch)
    for (pcm, bps, rate), data in zip(batch, got):
        ch = pcm.shape[1]
        fp, _ = ob.make_params(ch, bps, rate, level=5)
        o = ob.oracle_encode(fp, pcm)
        sizes = [len(f) for f in o["frames"]]
        want = ob.stream_head(4096, 4096, min(sizes), max(sizes), rate, ch, bps, len(pcm), o["md5"], is_last=False) + VC_BLOCK + b"".join(o["frames"])
        assert data == want
    # and back through the decoder object
    for (pcm, bps, rate), data in list(zip(batch, got))[:6]:
        d = stream.StreamDecoder(data, seekable=False)
        assert d.set_md5_checking(1) and d.init() == 0 and d.process_until_end_of_stream()
        assert np.array_equal(d.pcm(), pcm) and d.finish()
        d.close()
