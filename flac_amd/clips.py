"""Batch encode of many short independent clips (the 'thousands of 1-30 s clips' workload):
every clip becomes a complete FLAC stream, byte-identical to what the reference's encoder
writes for it.

Clips are grouped by (channels, bits per sample, sample rate).  The clips of a group are laid out
back to back in whole blocks, uploaded in ONE copy and coded by ONE call of
flacamd_encode_clips_device, which does everything per clip on the device: frame numbers from 0,
a short last block, the MD5 of the clip's samples (one lane per clip, beside the coding stages),
the frame size extremes, and the stream head in front of the clip's frames.  What comes back is one
buffer per group and the byte offsets of the streams in it; this module only slices it.
"""
import numpy as np

from . import engine

HEAD_BYTES = 86


def layout(lengths, blocksize):
    """Start of every clip in the shared buffer (whole blocks per clip) and the buffer's length, in wide samples."""
    nfr = -(-np.asarray(lengths, dtype=np.int64) // blocksize)
    ends = np.cumsum(nfr * blocksize)
    return (ends - nfr * blocksize), int(ends[-1]) if len(ends) else 0, nfr


def encode_group_device(enc, d_pcm, fmt, starts, lengths, nframes_total, slack=1.02, stream=None):
    """One format group resident on the device -> (uint8 device tensor of the streams back to back, offsets)."""
    import torch
    item = (2 if fmt == engine.PCM_S16 else 4) * enc.channels
    # frames rarely exceed the PCM they code (verbatim frames by a header); a batch that does says how much it needs
    cap = int(sum(int(n) for n in lengths) * item * slack) + nframes_total * 64 + HEAD_BYTES * len(lengths) + 4096
    d_out = torch.empty(cap, dtype=torch.uint8, device=d_pcm.device)
    try:
        offs = enc.encode_clips_device(d_pcm.data_ptr(), fmt, starts, lengths, d_out.data_ptr(), cap, stream)
    except engine.OutputTooSmall as big:
        cap = big.needed
        d_out = torch.empty(cap, dtype=torch.uint8, device=d_pcm.device)
        offs = enc.encode_clips_device(d_pcm.data_ptr(), fmt, starts, lengths, d_out.data_ptr(), cap, stream)
    return d_out, offs


def encode_groups_device(plans, device):
    """plans: list of (Encoder, d_pcm, fmt, starts, lengths, nframes).  The groups run side by side, each on a stream
    of its own from a thread of its own (the library call blocks, and the per-clip MD5 - a serial chain as long as
    the group's longest clip - would otherwise be waited for once per group).  Returns [(d_out, offsets)]."""
    import torch
    from concurrent.futures import ThreadPoolExecutor
    torch.cuda.synchronize(device)
    streams = [torch.cuda.Stream(device=device) for _ in plans]
    ready = None
    digests = [None] * len(plans)
    if len(plans) > 1:
        # the digests of all groups in one pass beside the coding stages: as long as the longest clip, once
        # (a stream of another priority: streams of one priority share the process's few hardware queues in turn, and a
        # group whose stream landed in the queue of this pass would wait behind it for as long as the longest clip)
        md5_stream = torch.cuda.Stream(device=device, priority=-1)
        digests = [torch.empty(len(p[3]) * 16, dtype=torch.uint8, device=p[1].device) for p in plans]
        engine.md5_groups_device(device, [(p[1].data_ptr(), p[2], p[0].channels, int(p[0].params.bits_per_sample), p[3], p[4], d.data_ptr())
                                          for p, d in zip(plans, digests)], stream=md5_stream.cuda_stream)
        ready = torch.cuda.Event()
        ready.record(md5_stream)

    def one(k):
        torch.cuda.set_device(device)
        enc, d_pcm, fmt, starts, lens, F = plans[k]
        if ready is not None:
            enc.use_digests(digests[k].data_ptr(), ready.cuda_event)
        return encode_group_device(enc, d_pcm, fmt, starts, lens, F, stream=streams[k].cuda_stream)

    if len(plans) == 1:
        return [one(0)]
    with ThreadPoolExecutor(max_workers=min(8, len(plans))) as pool:
        return list(pool.map(one, range(len(plans))))


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
    torch.cuda.set_device(device)
    plans, keep = [], []
    for (ch, bps, rate), members in groups.items():
        params = engine.make_params(ch, bps, rate, level=level)
        p2 = engine.Params.from_buffer_copy(params)
        if engine.lib().flacamd_params_resolve(p2) != 0:
            raise engine.EngineError(-1, "flacamd_params_resolve")
        bs = int(p2.blocksize)
        fmt, dt = (engine.PCM_S16, np.int16) if bps <= 16 else (engine.PCM_S32, np.int32)
        lens = [clips[i][0].shape[0] for i in members]
        starts, total, nfr = layout(lens, bs)
        host = np.zeros((total, ch), dtype=dt)
        for i, st, n in zip(members, starts, lens):
            host[st:st + n] = clips[i][0]
        d_pcm = torch.from_numpy(host).to(dev)
        F = int(nfr.sum())
        plans.append((engine.Encoder(params, max_frames=F, device=device), d_pcm, fmt, starts, lens, F))
        keep.append(members)
    results = encode_groups_device(plans, device)
    for (enc, *_), members, (d_out, offs) in zip(plans, keep, results):
        enc.close()
        h = d_out[:int(offs[-1])].cpu().numpy()
        for k, i in enumerate(members):
            out[i] = h[int(offs[k]):int(offs[k + 1])].tobytes()
    return out
