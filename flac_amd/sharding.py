"""Frame-range sharding of ONE stream over the ranks of a torch.distributed job, and the
host-side stitching of what the ranks return.

Frames are independent units given (frame number, samples): warm-up samples are stored
verbatim in every subframe (stream_encoder.c:3633-3634), so a rank needs no halo and there
is no collective on the data path.  What is sequential in the reference and therefore done
here, on the host, after the ranks finish: the exclusive scan of frame sizes (byte offsets),
ordered concatenation, min/max frame size for STREAMINFO (stream_encoder.c:2423-2426).  The
loose mid/side decision (presets 1 and 4) depends on the previous frames of its decision
period (stream_encoder.c:2958-2967, period from :905-908), so range starts are aligned to
that period: every rank then begins with a period leader and needs no neighbour state.

The per-rank encode is injected (`encode_range`): on a GPU box it is
flac_amd.engine.Encoder.encode_host / encode_device of this rank's GPU.

Decoding shards by byte ranges of the frame data (second half of this file): a rank decodes the frames that
start in its range, finding the first one by sync code, header CRC-8 and frame CRC-16
(flac_amd.engine.Decoder.decode_range); rank 0 checks that the shares follow one another and concatenates the PCM.
"""
import numpy as np


def loose_mid_side_period(sample_rate, blocksize):
    """Frames per loose mid/side decision period (stream_encoder.c:905-908; FLAC__real is float)."""
    v = int(float(np.float32(sample_rate)) * 0.4 / float(np.float32(blocksize)) + 0.5)
    return max(v, 1)


def frame_ranges(nframes, world, align=1):
    """Contiguous [first, last) frame ranges, one per rank, sizes a multiple of `align` (except
    the tail); ranks past the end get empty ranges."""
    if world < 1 or align < 1:
        raise ValueError("world and align must be >= 1")
    per = -(-nframes // world)
    per = -(-per // align) * align if per else 0
    out = []
    for r in range(world):
        a = min(r * per, nframes)
        out.append((a, min(a + per, nframes)))
    return out


def stitch(sizes_per_rank, bytes_per_rank):
    """Ordered concatenation + exclusive scan + STREAMINFO frame-size extremes."""
    sizes = np.concatenate([np.asarray(s, dtype=np.uint64) for s in sizes_per_rank]) if sizes_per_rank else np.zeros(0, np.uint64)
    offsets = np.zeros(len(sizes) + 1, dtype=np.uint64)
    np.cumsum(sizes, out=offsets[1:])
    data = b"".join(bytes_per_rank)
    if len(data) != int(offsets[-1]):
        raise ValueError("frame sizes do not add up to the bytes received")
    return dict(data=data, sizes=sizes.astype(np.uint32), offsets=offsets,
                min_framesize=int(sizes.min()) if len(sizes) else 0, max_framesize=int(sizes.max()) if len(sizes) else 0)


def encode_sharded(encode_range, nsamples, blocksize, align=None, group=None, device="cpu"):
    """Encode one stream of `nsamples` wide samples on all ranks of `group`.

    encode_range(first_frame, first_sample, end_sample, final) -> list of frame bytes for that
    range (final = the range holds the stream's last block).  Returns the stitched stream
    (dict, see stitch) on rank 0 and None elsewhere.  Works over RCCL ("nccl", tensors on this
    rank's GPU) and gloo (CPU tensors)."""
    import torch
    import torch.distributed as dist

    if align is None:
        raise ValueError("align: pass loose_mid_side_period(sample_rate, blocksize) for loose mid/side streams (presets 1 and 4), 1 otherwise - "
                         "a range that starts inside a decision period would code other bytes than the reference")
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    nframes = -(-nsamples // blocksize)
    f0, f1 = frame_ranges(nframes, world, align)[rank]
    frames = encode_range(f0, f0 * blocksize, min(f1 * blocksize, nsamples), f1 == nframes) if f1 > f0 else []
    if len(frames) != f1 - f0:
        raise RuntimeError("rank %d: expected %d frames, got %d" % (rank, f1 - f0, len(frames)))
    sizes = np.array([len(f) for f in frames], dtype=np.int64)
    blob = b"".join(frames)
    if world == 1:
        return stitch([sizes], [blob])

    # 1) every rank learns every rank's (frame count, byte count); 2) padded byte gather to rank 0
    mine = torch.tensor([len(sizes), len(blob)], dtype=torch.int64, device=device)
    counts = [torch.zeros(2, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(counts, mine, group=group)
    counts = [(int(c[0]), int(c[1])) for c in counts]
    max_f = max(max(c[0] for c in counts), 1)
    max_b = max(max(c[1] for c in counts), 1)
    t_sizes = torch.zeros(max_f, dtype=torch.int64, device=device)
    t_sizes[:len(sizes)] = torch.from_numpy(sizes).to(device)
    t_bytes = torch.zeros(max_b, dtype=torch.uint8, device=device)
    if blob:
        t_bytes[:len(blob)] = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(device)
    g_sizes = [torch.zeros_like(t_sizes) for _ in range(world)] if rank == 0 else None
    g_bytes = [torch.zeros_like(t_bytes) for _ in range(world)] if rank == 0 else None
    dist.gather(t_sizes, g_sizes, dst=0, group=group)
    dist.gather(t_bytes, g_bytes, dst=0, group=group)
    if rank != 0:
        return None
    return stitch([g_sizes[r][:counts[r][0]].cpu().numpy() for r in range(world)],
                  [g_bytes[r][:counts[r][1]].cpu().numpy().tobytes() for r in range(world)])


# ---- decode: byte ranges --------------------------------------------------------------------------

def byte_ranges(nbytes, world):
    """Contiguous [first, last) byte ranges of the frame data, one per rank."""
    if world < 1:
        raise ValueError("world must be >= 1")
    per = -(-nbytes // world)
    return [(min(r * per, nbytes), min((r + 1) * per, nbytes)) for r in range(world)]


def stitch_decoded(parts, nbytes):
    """parts: per rank, in rank order, dict(first_pos, end_pos, frames, pcm) for the frames that start in the rank's
    byte range.  A frame carries no length, so a rank finds its first frame by itself (sync code, header CRC-8,
    frame CRC-16); here the shares are checked to follow one another exactly and to cover [0, nbytes)."""
    at = 0
    out = []
    for r, p in enumerate(parts):
        if p["frames"] == 0:
            continue
        if p["first_pos"] != at:
            raise ValueError("rank %d starts at byte %d, the stream is decoded up to byte %d" % (r, p["first_pos"], at))
        at = p["end_pos"]
        out.append(p["pcm"])
    if at != nbytes:
        raise ValueError("frames end at byte %d of %d" % (at, nbytes))
    ch = parts[0]["pcm"].shape[1] if parts else 0
    return np.concatenate(out, axis=0) if out else np.zeros((0, ch), np.int32)


def decode_sharded(decode_range, nbytes, channels, group=None, device="cpu"):
    """Decode the frame data of one stream (nbytes bytes, known to every rank) on all ranks of `group`.

    decode_range(first_byte, last_byte) -> dict(first_pos, end_pos, frames, pcm int32 (samples, channels)) for the frames
    that start in that byte range (flac_amd.engine.Decoder.decode_range on this rank's GPU).  Returns the PCM of the
    whole stream on rank 0, None elsewhere.  No collective on the data path; the gather of the results runs over
    RCCL ("nccl", tensors on this rank's GPU) or gloo."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    b0, b1 = byte_ranges(nbytes, world)[rank]
    part = decode_range(b0, b1)
    if world == 1:
        return stitch_decoded([part], nbytes)
    pcm = np.ascontiguousarray(part["pcm"], dtype=np.int32)
    mine = torch.tensor([part["first_pos"], part["end_pos"], part["frames"], pcm.shape[0]], dtype=torch.int64, device=device)
    metas = [torch.zeros(4, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(metas, mine, group=group)
    metas = [[int(v) for v in m] for m in metas]
    max_s = max(max(m[3] for m in metas), 1)
    t = torch.zeros((max_s, channels), dtype=torch.int32, device=device)
    if pcm.shape[0]:
        t[:pcm.shape[0]] = torch.from_numpy(pcm).to(device)
    got = [torch.zeros_like(t) for _ in range(world)] if rank == 0 else None
    dist.gather(t, got, dst=0, group=group)
    if rank != 0:
        return None
    return stitch_decoded([dict(first_pos=m[0], end_pos=m[1], frames=m[2], pcm=got[r][:m[3]].cpu().numpy()) for r, m in enumerate(metas)], nbytes)
