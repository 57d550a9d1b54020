"""Batch encode of many short independent clips (the 'thousands of 1-30 s clips' workload):
every clip becomes a complete FLAC stream, byte-identical to what the reference's encoder
writes for it.

Clips are grouped by (channels, bits per sample, sample rate).  The clips of a group are laid out
back to back in whole blocks, uploaded in ONE copy and coded by ONE engine call
(flacamd_encode_streams_device: frame numbers restart at 0 for every clip, every clip's last block
may be short); the MD5 of every clip of the group comes from one launch of the on-device MD5 kernel
(one lane per clip) on a stream of its own, so the serial-per-stream digest overlaps the encode.
"""
import numpy as np

from . import engine


def encode_clips(clips, level=5, device=0):
    """clips: list of (pcm, bits_per_sample, sample_rate) with pcm an int array (samples, channels).
    Returns a list of bytes objects, one complete stream per clip, in input order."""
    import torch
    from concurrent.futures import ThreadPoolExecutor
    groups = {}
    for idx, (pcm, bps, rate) in enumerate(clips):
        pcm = np.asarray(pcm)
        if pcm.ndim != 2 or pcm.shape[0] == 0:
            raise ValueError("clip %d: pcm must be (samples, channels) and not empty" % idx)
        groups.setdefault((pcm.shape[1], bps, rate), []).append(idx)
    out = [None] * len(clips)
    dev = torch.device("cuda", device)
    torch.cuda.set_device(device)
    plans = []
    pool = ThreadPoolExecutor(max_workers=max(1, min(8, len(groups))))
    # ---- phase 1: upload every group, start its MD5 launch on a stream of its own (one lane per clip:
    # the digests take as long as the longest clip, which the encode calls below overlap)
    for (ch, bps, rate), members in groups.items():
        params = engine.make_params(ch, bps, rate, level=level)
        p2 = engine.Params.from_buffer_copy(params)
        if engine.lib().flacamd_params_resolve(p2) != 0:
            raise engine.EngineError(-1, "flacamd_params_resolve")
        bs = int(p2.blocksize)
        fmt, dt = (engine.PCM_S16, np.int16) if bps <= 16 else (engine.PCM_S32, np.int32)
        item = np.dtype(dt).itemsize * ch
        lens = [clips[i][0].shape[0] for i in members]
        # every clip starts on a block boundary of the shared buffer: frame f of the batch is samples [f * bs, f * bs + bs)
        nfr = [-(-n // bs) for n in lens]
        starts, pos = [], 0
        for f in nfr:
            starts.append(pos)
            pos += f * bs
        host = np.zeros((pos, ch), dtype=dt)
        for i, st, n in zip(members, starts, lens):
            host[st:st + n] = clips[i][0]
        d_pcm = torch.from_numpy(host).to(dev)
        offs = []   # the padding between clips is hashed as streams of its own and dropped
        for st, n in zip(starts, lens):
            offs += [st, st + n]
        offs.append(pos)
        side = torch.cuda.Stream(device=dev)
        fut = pool.submit(engine.md5_device, d_pcm.data_ptr(), fmt, ch, bps, offs, device, side.cuda_stream)
        plans.append((ch, bps, rate, members, params, bs, fmt, lens, starts, nfr, d_pcm, fut, side))
    # ---- phase 2: one engine call per group on the resident buffer; outputs stay on the device until the call returns
    results = []
    for ch, bps, rate, members, params, bs, fmt, lens, starts, nfr, d_pcm, fut, side in plans:
        total_frames = sum(nfr)
        enc = engine.Encoder(params, max_frames=total_frames, device=device)
        cap = enc.max_output_bytes(total_frames)
        d_out = torch.empty(cap, dtype=torch.uint8, device=dev)
        d_fb = torch.empty(total_frames, dtype=torch.int32, device=dev)
        try:
            total = enc.encode_streams_device(d_pcm.data_ptr(), fmt, starts, lens, d_out.data_ptr(), cap, d_fb.data_ptr())
        except engine.OutputTooSmall as big:   # frames beyond the verbatim bound (the reference writes them too)
            cap = big.needed
            d_out = torch.empty(cap, dtype=torch.uint8, device=dev)
            total = enc.encode_streams_device(d_pcm.data_ptr(), fmt, starts, lens, d_out.data_ptr(), cap, d_fb.data_ptr())
        enc.close()
        results.append((d_out[:total].cpu().numpy(), d_fb.cpu().numpy()))
    # ---- phase 3: digests, stream heads, assembly
    for (ch, bps, rate, members, params, bs, fmt, lens, starts, nfr, d_pcm, fut, side), (h_out, h_fb) in zip(plans, results):
        digests = fut.result()[::2]
        o_off = f_off = 0
        for k, i in enumerate(members):
            sizes = h_fb[f_off:f_off + nfr[k]]
            nbytes = int(sizes.sum())
            head = engine.stream_head(bs, int(sizes.min()), int(sizes.max()), rate, ch, bps, lens[k], digests[k])
            out[i] = head + h_out[o_off:o_off + nbytes].tobytes()
            o_off += nbytes
            f_off += nfr[k]
    pool.shutdown()
    return out
