"""Batch encode of many short independent clips (the 'thousands of 1-30 s clips' workload):
every clip becomes a complete FLAC stream, byte-identical to what the reference's encoder
writes for it.

Clips are grouped by (channels, bits per sample, sample rate): one engine encoder per group,
sized for the group's longest clip and reused; the clips of a group are uploaded in ONE copy,
back to back in device memory; each clip is one engine call on that resident buffer (frames of
one clip are a frame range like any other: frame numbers restart at 0, the last block may be
short); the MD5 of every clip of the group comes from one launch of the on-device MD5 kernel
(one lane per clip), so the serial-per-stream digest does not serialise the batch.
"""
import numpy as np

from . import engine


def encode_clips(clips, level=5, device=0):
    """clips: list of (pcm, bits_per_sample, sample_rate) with pcm an int array (samples, channels).
    Returns a list of bytes objects, one complete stream per clip, in input order."""
    import torch
    groups = {}
    for idx, (pcm, bps, rate) in enumerate(clips):
        pcm = np.asarray(pcm)
        if pcm.ndim != 2 or pcm.shape[0] == 0:
            raise ValueError("clip %d: pcm must be (samples, channels) and not empty" % idx)
        groups.setdefault((pcm.shape[1], bps, rate), []).append(idx)
    out = [None] * len(clips)
    dev = torch.device("cuda", device)
    for (ch, bps, rate), members in groups.items():
        params = engine.make_params(ch, bps, rate, level=level)
        probe = engine.lib()
        p2 = engine.Params.from_buffer_copy(params)
        if probe.flacamd_params_resolve(p2) != 0:
            raise engine.EngineError(-1, "flacamd_params_resolve")
        bs = int(p2.blocksize)
        fmt, dt = (engine.PCM_S16, np.int16) if bps <= 16 else (engine.PCM_S32, np.int32)
        lens = [clips[i][0].shape[0] for i in members]
        # every clip starts on a 16-byte boundary of the shared buffer: the vector-load kernels require it
        align = 16 // (np.dtype(dt).itemsize * ch) if 16 % (np.dtype(dt).itemsize * ch) == 0 else 16
        starts, pos = [], 0
        for n in lens:
            starts.append(pos)
            pos += -(-n // align) * align
        host = np.zeros((pos, ch), dtype=dt)
        for i, st, n in zip(members, starts, lens):
            host[st:st + n] = np.asarray(clips[i][0]).astype(dt)
        d_pcm = torch.from_numpy(host).to(dev)
        max_frames = max(-(-n // bs) for n in lens)
        enc = engine.Encoder(params, max_frames=max_frames, device=device)
        cap = enc.max_output_bytes(max_frames)
        d_out = torch.empty(cap, dtype=torch.uint8, device=dev)
        d_fb = torch.empty(max_frames, dtype=torch.int32, device=dev)
        # MD5 of all clips of the group in one launch: offsets are wide-sample positions
        # (the alignment gaps between clips are hashed as streams of their own and dropped)
        offs = []
        for st, n in zip(starts, lens):
            offs += [st, st + n]
        digests = engine.md5_device(d_pcm.data_ptr(), fmt, ch, bps, offs, device=device)[::2]
        for k, (i, st, n) in enumerate(zip(members, starts, lens)):
            nfr = -(-n // bs)
            total = enc.encode_device(d_pcm.data_ptr() + st * ch * np.dtype(dt).itemsize, fmt, n, d_out.data_ptr(), cap, d_fb.data_ptr(), final=True, first_frame=0)
            data = d_out[:total].cpu().numpy().tobytes()
            sizes = d_fb[:nfr].cpu().numpy()
            head = engine.stream_head(bs, int(sizes.min()), int(sizes.max()), rate, ch, bps, n, digests[k])
            out[i] = head + data
        enc.close()
    return out
