"""GPU tests of flacamd_encode_streams_device: several complete streams coded in one pass must give,
stream by stream, exactly the frames the engine (and so the reference) gives for each stream alone."""
import numpy as np
import pytest
import torch

from flac_amd import engine, synth

pytestmark = pytest.mark.gpu


def _alone(params, pcm, fmt_dtype):
    enc = engine.Encoder(params, max_frames=64)
    frames, _ = enc.encode_host(np.ascontiguousarray(pcm.astype(fmt_dtype)), final=True, first_frame=0)
    enc.close()
    return frames


@pytest.mark.parametrize("ch,bps,level,kw", [
    (2, 16, 8, {}),                     # register-resident kernels for the full blocks, general ones for the short tails
    (2, 16, 1, {}),                     # loose mid/side: decision periods restart with every stream
    (2, 16, 4, dict(blocksize=1152)),
    (1, 16, 5, {}),
    (2, 24, 5, {}),
    (3, 24, 8, dict(blocksize=576)),    # general kernels only
])
def test_streams_in_one_pass_equal_streams_alone(gpu_engine, ch, bps, level, kw):
    rng = np.random.default_rng(5)
    params = engine.make_params(ch, bps, 44100, level=level, **kw)
    p2 = engine.Params.from_buffer_copy(params)
    assert engine.lib().flacamd_params_resolve(p2) == 0
    bs = int(p2.blocksize)
    dt = np.int16 if bps <= 16 else np.int32
    lens = [int(x) for x in [bs * 3, bs * 2 + 17, 5, bs, bs * 4 + bs // 2, 1, bs - 1, bs * 9 + 100]]
    streams = []
    for i, n in enumerate(lens):
        if bps <= 16:
            base = synth.music_like(n / 44100.0 + 0.01, seed=40 + i)[:n]
            pcm = np.stack([base[:, c % 2] for c in range(ch)], axis=1)
        else:
            pcm = synth.multichannel_24(n / 96000.0 + 0.01, channels=ch, seed=60 + i)[:n]
        if i == 3:
            pcm = np.zeros_like(pcm) + 7      # a constant stream in the middle
        streams.append(np.ascontiguousarray(pcm))
    nfr = [-(-n // bs) for n in lens]
    starts = np.cumsum([0] + [f * bs for f in nfr])[:-1]
    host = rng.integers(-3, 3, size=(sum(nfr) * bs, ch)).astype(dt)   # padding is never read as audio: make it non-zero
    for st, pcm in zip(starts, streams):
        host[st:st + len(pcm)] = pcm
    d_pcm = torch.from_numpy(host).cuda()
    F = sum(nfr)
    enc = engine.Encoder(params, max_frames=F)
    cap = enc.max_output_bytes(F)
    d_out = torch.empty(cap, dtype=torch.uint8, device="cuda")
    d_fb = torch.empty(F, dtype=torch.int32, device="cuda")
    fmt = engine.PCM_S16 if bps <= 16 else engine.PCM_S32
    total = enc.encode_streams_device(d_pcm.data_ptr(), fmt, starts, lens, d_out.data_ptr(), cap, d_fb.data_ptr())
    sizes = d_fb.cpu().numpy()
    data = d_out[:total].cpu().numpy().tobytes()
    assert int(sizes.sum()) == total
    off = f = 0
    for pcm, k in zip(streams, nfr):
        want = _alone(params, pcm, dt)
        got = []
        for j in range(k):
            got.append(data[off:off + int(sizes[f + j])])
            off += int(sizes[f + j])
        f += k
        assert got == want
    # bad layouts are refused: a gap, a stream that does not start where the previous one's blocks end
    with pytest.raises(engine.EngineError):
        enc.encode_streams_device(d_pcm.data_ptr(), fmt, [0, bs * 4], [bs * 3, bs], d_out.data_ptr(), cap, d_fb.data_ptr())
    with pytest.raises(engine.EngineError):
        enc.encode_streams_device(d_pcm.data_ptr(), fmt, [0], [0], d_out.data_ptr(), cap, d_fb.data_ptr())
    enc.close()


def test_a_frame_beyond_the_verbatim_size_inside_a_batch(gpu_engine):
    """One of the streams of the batch holds a frame of about twice the verbatim size (full-scale 24-bit noise: a narrow
    partition sum of the reference wraps, stream_encoder.c:3808): the call names the output size it takes, and the
    frames of all streams equal the oracle's."""
    import oracle_bindings as ob
    kw = dict(level=8, blocksize=1000, max_residual_partition_order=4)
    fp, _ = ob.make_params(2, 24, 48000, **kw)
    params = engine.make_params(2, 24, 48000, **kw)
    rng = np.random.default_rng(1)
    streams = [rng.integers(-20000, 20000, (2500, 2)).astype(np.int32),
               np.random.default_rng(3).integers(-7300000, 7300000, (3000, 2)).astype(np.int32),
               rng.integers(-2000, 2000, (1000, 2)).astype(np.int32)]
    want = []
    for pcm in streams:
        want += ob.oracle_encode(fp, pcm)["frames"]
    assert max(len(f) for f in want) > 9000
    lens = [len(s) for s in streams]
    nfr = [-(-n // 1000) for n in lens]
    starts = np.cumsum([0] + [f * 1000 for f in nfr])[:-1]
    host = np.zeros((sum(nfr) * 1000, 2), np.int32)
    for st, pcm in zip(starts, streams):
        host[st:st + len(pcm)] = pcm
    d_pcm = torch.from_numpy(host).cuda()
    F = sum(nfr)
    enc = engine.Encoder(params, max_frames=F)
    cap = 8 * 1024      # far too small on purpose
    d_out = torch.empty(cap, dtype=torch.uint8, device="cuda")
    d_fb = torch.empty(F, dtype=torch.int32, device="cuda")
    with pytest.raises(engine.OutputTooSmall) as big:
        enc.encode_streams_device(d_pcm.data_ptr(), engine.PCM_S32, starts, lens, d_out.data_ptr(), cap, d_fb.data_ptr())
    cap = big.value.needed
    assert cap == sum(len(f) for f in want)
    d_out = torch.empty(cap, dtype=torch.uint8, device="cuda")
    total = enc.encode_streams_device(d_pcm.data_ptr(), engine.PCM_S32, starts, lens, d_out.data_ptr(), cap, d_fb.data_ptr())
    enc.close()
    sizes = d_fb.cpu().numpy()
    data = d_out[:total].cpu().numpy().tobytes()
    got, off = [], 0
    for s in sizes:
        got.append(data[off:off + int(s)])
        off += int(s)
    assert got == want
