"""The secondary BASELINE configurations at their stated sizes, through size-independent properties
(encode -> device decode round trips, frames sampled against the oracle under their own frame
numbers, per-clip MD5 against hashlib, stitched shards equal to the whole)."""
import hashlib

import numpy as np
import pytest

import oracle_bindings as ob
from flac_amd import clips, sharding, synth

pytestmark = pytest.mark.gpu
N = 4096


def test_c3_ten_minutes_8ch_24bit_exhaustive(gpu_engine):
    """flac -8 -e -p on 600 s of 8-channel 24-bit / 96 kHz: the whole batch decodes on the device to the PCM it was made
    from, and frames sampled across it equal the oracle's coding of the same block under the same frame number."""
    import torch
    base = synth.multichannel_24(10.0, rate=96000, channels=8, seed=4321)
    reps = 60
    n = len(base) * reps
    F = (n + N - 1) // N
    kw = dict(level=8, do_exhaustive_model_search=1, do_qlp_coeff_prec_search=1)
    enc = gpu_engine.Encoder(gpu_engine.make_params(8, 24, 96000, **kw), max_frames=F)
    dev = torch.device("cuda", 0)
    d_pcm = torch.from_numpy(base).to(dev).repeat(reps, 1)
    cap = n * 32 + F * 256
    d_out = torch.empty(cap, dtype=torch.uint8, device=dev)
    d_fb = torch.empty(F, dtype=torch.int32, device=dev)
    total = enc.encode_device(d_pcm.data_ptr(), gpu_engine.PCM_S32, n, d_out.data_ptr(), cap, d_fb.data_ptr())
    enc.close()
    sizes = d_fb.cpu().numpy().astype(np.int64)
    assert F == 14063 and int(sizes.sum()) == total
    offs = np.concatenate([[0], np.cumsum(sizes)])
    fp, _ = ob.make_params(8, 24, 96000, **kw)
    for f in (0, 1, 2345, 7031, F - 2, F - 1):
        a, b = f * N, min((f + 1) * N, n)
        blk = np.ascontiguousarray(d_pcm[a:b].cpu().numpy())
        want = ob.oracle_encode(fp, blk, first_frame=f, final=(f == F - 1))["frames"][0]
        got = d_out[int(offs[f]):int(offs[f + 1])].cpu().numpy().tobytes()
        assert got == want, "frame %d" % f
    dec = gpu_engine.Decoder(8, 24, N, F + 256, total + 64)
    d_back = torch.empty((n, 8), dtype=torch.int32, device=dev)
    info = dec.decode_device(d_out.data_ptr(), total, d_back.data_ptr(), gpu_engine.PCM_S32, n)
    dec.close()
    assert info["frames"] == F and info["samples"] == n and info["bytes"] == total
    assert torch.equal(d_back, d_pcm)


def test_c4_multi_gigabyte_stream_decodes_whole_and_in_byte_ranges(gpu_engine):
    """A 2.2 GB -8 stream (5 h of 16/44.1 stereo): decoded in one pass and, as 8 byte ranges the way 8 GPUs would share
    it (every range finds its own first frame by sync code + CRC-8 + CRC-16), both bit-identical to the PCM."""
    import torch
    dev = torch.device("cuda", 0)
    base = torch.from_numpy(synth.music_like(60, seed=1234).astype(np.int16)).to(dev)
    d_pcm = base.repeat(300, 1)
    n = d_pcm.shape[0]
    F = (n + N - 1) // N
    enc = gpu_engine.Encoder(gpu_engine.make_params(2, 16, 44100, level=8), max_frames=F)
    cap = n * 4 + F * 64
    d_out = torch.empty(cap + 64, dtype=torch.uint8, device=dev)
    d_fb = torch.empty(F, dtype=torch.int32, device=dev)
    total = enc.encode_device(d_pcm.data_ptr(), gpu_engine.PCM_S16, n, d_out.data_ptr(), cap, d_fb.data_ptr())
    enc.close()
    assert total > 2_000_000_000 and F == 193799
    ends = torch.cumsum(d_fb.to(torch.int64), 0).cpu().numpy()
    d_back = torch.empty((n, 2), dtype=torch.int16, device=dev)
    dec = gpu_engine.Decoder(2, 16, N, int(F * 1.25) + 64, total + 64)
    info = dec.decode_device(d_out.data_ptr(), total, d_back.data_ptr(), gpu_engine.PCM_S16, n)
    assert info == dict(frames=F, samples=n, first_number=0, bytes=total)
    assert torch.equal(d_back, d_pcm)
    # 8 byte ranges: the frames that START in a range belong to it; a range is decoded from its first frame on
    d_back.zero_()
    dec.set_tolerant(True)
    world = 8
    frames_seen, at = 0, 0
    starts = np.concatenate([[0], ends])
    for r, (b0, b1) in enumerate(sharding.byte_ranges(total, world)):
        f_lo = int(np.searchsorted(starts[:-1], b0, side="left"))      # first frame starting at or after b0
        f_hi = int(np.searchsorted(starts[:-1], b1, side="left"))
        if f_hi == f_lo:
            continue
        # the window a rank decodes: from a 16-byte aligned position at or before its range to the end of its last frame
        w0 = b0 & ~15
        w1 = int(starts[f_hi])
        rinfo = dec.decode_device(d_out.data_ptr() + w0, w1 - w0, d_back.data_ptr() + (f_lo * N) * 4, gpu_engine.PCM_S16, n - f_lo * N)
        t = dec.candidate_table(int(F * 1.25) + 64)
        good = (t["info"] & 3) == 3
        first = int(t["pos"][good][np.searchsorted(t["pos"][good], b0 - w0)]) + w0
        assert first == int(starts[f_lo]), "range %d finds its first frame" % r
        # frames decoded from the bytes in front of the range's first frame (there may be a whole frame there) are not its share
        lead = int((t["pos"][good] < (b0 - w0)).sum())
        assert rinfo["frames"] - lead == f_hi - f_lo
        frames_seen += f_hi - f_lo
        at = w1
    dec.close()
    assert frames_seen == F and at == total


def test_c5_ten_thousand_clips_complete_streams(gpu_engine):
    """10 000 clips (1-30 s, 16/24-bit, 44.1/48 kHz, mono/stereo, seed 777) at -5 as complete streams assembled on the
    device: every stream starts with the head the reference writes, STREAMINFO carries the clip's length, frame size
    extremes and the MD5 of its samples (hashlib on a sample of clips), a sample of clips is byte-identical to the
    oracle's stream, and one whole group decodes on the device to the PCM it was made from."""
    import torch
    dev = torch.device("cuda", 0)
    nclips = 10000
    rng = np.random.default_rng(777)
    dur = rng.uniform(1.0, 30.0, nclips)
    bps = np.where(rng.random(nclips) < 0.5, 16, 24)
    rate = np.where(rng.random(nclips) < 0.5, 44100, 48000)
    ch = np.where(rng.random(nclips) < 0.5, 1, 2)
    plans, hosts = [], []
    for c, b, r in sorted({(int(c), int(b), int(r)) for c, b, r in zip(ch, bps, rate)}):
        idx = np.nonzero((ch == c) & (bps == b) & (rate == r))[0]
        if b == 16:
            progs = [np.ascontiguousarray(synth.music_like(30.0, rate=r, seed=5 + k)[:, :c].astype(np.int16)) for k in range(4)]
            fmt = gpu_engine.PCM_S16
        else:
            progs = [np.ascontiguousarray(synth.multichannel_24(30.0, rate=r, channels=c, seed=9 + k)) for k in range(4)]
            fmt = gpu_engine.PCM_S32
        d_progs = [torch.from_numpy(p).to(dev) for p in progs]
        lens = (dur[idx] * r).astype(np.int64)
        starts, total, nfr = clips.layout(lens, N)
        d_pcm = torch.zeros((total, c), dtype=d_progs[0].dtype, device=dev)
        for j, (st, ln) in enumerate(zip(starts, lens)):
            d_pcm[int(st):int(st) + int(ln)] = d_progs[j % 4][:int(ln)]
        Fg = int(nfr.sum())
        enc = gpu_engine.Encoder(gpu_engine.make_params(c, b, r, level=5), max_frames=Fg)
        plans.append((enc, d_pcm, fmt, starts, lens, Fg))
        hosts.append((c, b, r, progs, nfr))
    outs = clips.encode_groups_device(plans, 0)
    seen = 0
    for (enc, d_pcm, fmt, starts, lens, Fg), (c, b, r, progs, nfr), (d_out, offs) in zip(plans, hosts, outs):
        enc.close()
        nb = (b + 7) // 8
        offs = offs.astype(np.int64)
        assert len(offs) == len(lens) + 1 and offs[0] == 0
        heads = torch.stack([d_out[int(o):int(o) + 86] for o in offs[:-1][:: max(1, len(lens) // 200)]]).cpu().numpy()
        assert (heads[:, :4] == np.frombuffer(b"fLaC", np.uint8)).all()
        fp, _ = ob.make_params(c, b, r, level=5)
        for j in list(range(0, len(lens), max(1, len(lens) // 6)))[:6]:
            stream = d_out[int(offs[j]):int(offs[j + 1])].cpu().numpy().tobytes()
            pcm = progs[j % 4][:int(lens[j])]
            want = ob.oracle_encode(fp, pcm.astype(np.int32))
            sizes = [len(f) for f in want["frames"]]
            md5 = hashlib.md5(pcm.astype("<i4").reshape(-1).view(np.uint8).reshape(-1, 4)[:, :nb].tobytes()).digest()
            head = ob.stream_head(N, N, min(sizes), max(sizes), r, c, b, len(pcm), md5, is_last=False)
            assert stream[:42] == head and stream[86:] == b"".join(want["frames"]), (c, b, r, j)
        seen += len(lens)
    assert seen == nclips
    # one whole group through the device decoder: strip the heads by decoding every stream's frames in place
    enc, d_pcm, fmt, starts, lens, Fg = plans[0]
    d_out, offs = outs[0]
    c, b, r, progs, nfr = hosts[0]
    offs = offs.astype(np.int64)
    dec = gpu_engine.Decoder(c, b, N, 512, 8 << 20)
    for j in range(0, len(lens), max(1, len(lens) // 40)):
        a0 = int(offs[j]) + 86
        w0 = a0 & ~15
        nbytes = int(offs[j + 1]) - w0
        back = torch.empty((int(lens[j]) + N, c), dtype=torch.int16 if fmt == gpu_engine.PCM_S16 else torch.int32, device=dev)
        dec.set_tolerant(True)   # the window starts up to 15 bytes ahead of the first frame
        info = dec.decode_device(d_out.data_ptr() + w0, nbytes, back.data_ptr(), fmt, int(lens[j]) + N)
        assert info["samples"] == int(lens[j]) and info["frames"] == int(nfr[j])
        assert torch.equal(back[:int(lens[j])], d_pcm[int(starts[j]):int(starts[j]) + int(lens[j])])
    dec.close()
