#!/usr/bin/env python3
"""Headline benchmark: FLAC -8 encode of 16-bit/44.1 kHz stereo on MI355X.

One "step" = one pass of the whole hot path (prep -> autocorrelation -> model -> search ->
pack -> gather) over one batch: one hour of synthetic audio (38 760 frames of 4096 samples,
158.76 M wide samples) already resident in HBM as native-width int16, producing the packed
FLAC frames and their sizes in HBM.  With N GPUs every rank encodes its own hour-long frame
range of one N-hour stream (frames are independent units: no data-path collective; the host
would concatenate), so scaling is weak.

Prints ONE JSON line (rank 0).  `value` is whole-job M wide-samples/s.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FRAME = 4096
RATE = 44100
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak (spec)
QUEUE_DEPTH = 16   # FLACAMD_QUEUE_DEPTH: batches one encoder may have outstanding


# kernel that runs each timed stage for this workload (16-bit stereo, blocksize 4096)
KERNEL_OF_STAGE = {"autoc_beside_prep": "fa_autoc_ms16_kernel<13, 1>", "prep_join_wait": "fa_prep_stereo16_lean_kernel", "model": "fa_model_kernel<12>",
                   "search": "fa_search_fast_kernel<unsigned int, 1, 1, 17>", "pack": "fa_pack_fast_kernel<2, true, true, true>", "gather": "fa_scan_kernel"}


PMC_SUMMARY = "profiles/r3_pmc_summary.json"


def pmc_traffic(kernel, minutes, stage_ms):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes of this same
    command (PMC_SUMMARY: FETCH_SIZE and WRITE_SIZE collected in separate runs, FETCH_SIZE doubled
    as measured for gfx950).  The counters cannot be collected inside a timed run, so the figure
    is only reported while it still describes the kernel that ran: (None, why) when no pass covers
    the kernel or when the profiled average duration differs from this run's stage time by more
    than 10 % (the kernel has changed since the passes were made)."""
    try:
        with open(os.path.join(ROOT, PMC_SUMMARY)) as f:
            summ = json.load(f)
        if summ.get("minutes") != minutes:
            return None, "the committed passes cover %s minutes per step, this run %s" % (summ.get("minutes"), minutes)
        k = summ["kernels"].get(kernel)
        if k is None:
            return None, "no committed pass covers this kernel"
        avg = k.get("avg_ms")
        if avg is None or stage_ms <= 0 or abs(avg - stage_ms) / stage_ms > 0.10:
            return None, "stale: profiled average %s ms against %.4f ms in this run" % (avg, stage_ms)
        return int(k["hbm_bytes_per_launch"]), "committed rocprofv3 PMC passes of this command (%s, kernel average %.4f ms there, %.4f ms here), not measured in this run" % (PMC_SUMMARY, avg, stage_ms)
    except (OSError, ValueError, KeyError) as ex:
        return None, "no usable summary: %s" % ex


DISTINCT_MINUTES = 10


def build_workload(seconds_base=60, repeats=60):
    """One hour = ten different 60 s seeded synthetic programmes (6460 distinct frames) laid end to end and tiled six
    times (int16 stereo).  Returns the first programme (what the CPU baselines and the API client work on) and the hour."""
    from flac_amd import synth
    progs = [synth.music_like(seconds_base, rate=RATE, seed=1234 + k).astype(np.int16) for k in range(min(DISTINCT_MINUTES, repeats))]
    block = np.concatenate(progs, axis=0)
    whole, rest = divmod(repeats, len(progs))
    pcm = np.concatenate([np.tile(block, (whole, 1))] + progs[:rest], axis=0)
    return progs[0], pcm


def cpu_baseline(base_pcm, budget_s=12.0):
    """Time the reference encoder (oracle/_ref, compiled from the reference's own sources) on
    the host cores, on a bounded sample: repeated passes over the 60 s base programme."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_bindings as ob
    pcm32 = np.ascontiguousarray(base_pcm.astype(np.int32))
    n = pcm32.shape[0]
    if ob.have_ref():
        _, rp = ob.make_params(2, 16, RATE, level=8)
        fn = lambda: ob.ref_encode(rp, pcm32)
        kind = "reference"
    else:
        fp, _ = ob.make_params(2, 16, RATE, level=8)
        fn = lambda: ob.oracle_encode(fp, pcm32)
        kind = "port"
    first = fn()  # warm caches
    ref_bytes = sum(len(f) for f in first["frames"])
    t0 = time.perf_counter()
    passes = 0
    best = 1e30
    while True:
        t1 = time.perf_counter()
        fn()
        dt = time.perf_counter() - t1
        best = min(best, dt)
        passes += 1
        if time.perf_counter() - t0 > budget_s or passes >= 24:
            break
    single = n / best / 1e6
    out = {"value": round(single, 3), "unit": "MSamples/s", "cores": 1, "kind": kind, "ref_bytes_60s": ref_bytes,
           "sample": "%d passes of flac -8 over a 60 s 16/44.1 stereo programme (%d wide samples), best pass, single stream" % (passes, n)}
    # all host cores: one process per core of this job's CPU share
    try:
        import multiprocessing as mp
        # one process per core this job may use: the box sets OMP_NUM_THREADS to its CPU share (16 for one GPU) while
        # os.cpu_count() reports the whole machine; both are recorded
        try:
            affinity = len(os.sched_getaffinity(0))
        except (AttributeError, OSError):
            affinity = os.cpu_count() or 1
        share = int(os.environ.get("OMP_NUM_THREADS", "0")) or affinity
        cores = max(1, min(share, affinity))
        ctx = mp.get_context("fork")
        reps = 3
        t1 = time.perf_counter()
        procs = [ctx.Process(target=_cpu_worker, args=(kind, reps)) for _ in range(cores)]
        for p in procs:
            p.start()
        for p in procs:
            p.join()
        dt = time.perf_counter() - t1
        out["all_cores"] = {"value": round(cores * reps * n / dt / 1e6, 3), "cores": cores, "host_cpu_count": os.cpu_count(), "affinity": affinity,
                            "cpu_share_env": os.environ.get("OMP_NUM_THREADS"),
                            "sample": "%d processes x %d passes of the same programme, wall clock incl. process start" % (cores, reps)}
    except Exception as ex:  # the single-stream figure stands on its own
        out["all_cores"] = {"error": str(ex)}
    return out


def _wav_bytes(pcm16, rate):
    import struct
    data = np.ascontiguousarray(pcm16.astype("<i2")).tobytes()
    ch = pcm16.shape[1]
    return b"RIFF" + struct.pack("<I", 36 + len(data)) + b"WAVEfmt " + struct.pack("<IHHIIHH", 16, 1, ch, rate, rate * ch * 2, ch * 2, 16) + b"data" + struct.pack("<I", len(data)) + data


def _raw24(pcm32):
    return np.ascontiguousarray(pcm32.astype("<i4")).reshape(-1, 1).view(np.uint8).reshape(-1, 4)[:, :3].tobytes()


def cpu_tool_baselines(base_pcm, clips=24):
    """The reference's own command line tool (oracle/_ref/flac, built from the reference's sources) on files in /dev/shm
    (page cache), single stream, best wall time: `flac -s -f -8` on the 60 s programme of the headline workload, and a
    bounded sample of each other configuration: `-8 -e -p` on 10 s of the 8-channel 24-bit/96 kHz programme, `flac -d` of
    the -8 file, `-5` on the first `clips` clips of the clip workload's draw."""
    import shutil
    import subprocess
    import tempfile
    exe = os.path.join(ROOT, "oracle", "_ref", "flac")
    if not os.path.exists(exe):
        return {"error": "oracle/_ref/flac not built"}
    from flac_amd import synth
    tmp = tempfile.mkdtemp(prefix="flacamd_cpu_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)

    def best(cmd, reps):
        b = 1e30
        for _ in range(reps):
            t0 = time.perf_counter()
            r = subprocess.run(cmd, cwd=tmp, capture_output=True, text=True)
            dt = time.perf_counter() - t0
            if r.returncode != 0:
                raise RuntimeError("%s: %s" % (" ".join(cmd), r.stderr[-200:]))
            b = min(b, dt)
        return b

    out = {}
    try:
        n = base_pcm.shape[0]
        with open(os.path.join(tmp, "c2.wav"), "wb") as f:
            f.write(_wav_bytes(base_pcm, RATE))
        dt = best([exe, "-s", "-f", "-8", "-o", "c2.flac", "c2.wav"], 3)
        out["c2_tool"] = {"value": round(n / dt / 1e6, 3), "unit": "MSamples/s", "cores": 1, "kind": "reference", "seconds": round(dt, 3),
                          "sample": "flac -s -f -8 on the 60 s 16/44.1 stereo programme as a WAVE file in /dev/shm, best of 3 (MD5 on, file written)"}
        dt = best([exe, "-s", "-f", "-d", "-o", "c4.wav", "c2.flac"], 3)
        out["c4_decode"] = {"value": round(n / dt / 1e6, 3), "unit": "MSamples/s", "cores": 1, "kind": "reference", "seconds": round(dt, 3),
                            "sample": "flac -s -f -d of that file's -8 stream (MD5 checked, WAVE written to /dev/shm), best of 3"}
        c3 = synth.multichannel_24(10.0, rate=96000, channels=8, seed=4321)
        with open(os.path.join(tmp, "c3.raw"), "wb") as f:
            f.write(_raw24(c3))
        dt = best([exe, "-s", "-f", "-8", "-e", "-p", "--force-raw-format", "--endian=little", "--sign=signed", "--channels=8", "--bps=24", "--sample-rate=96000",
                   "-o", "c3.flac", "c3.raw"], 1)
        out["c3"] = {"value": round(len(c3) / dt / 1e6, 4), "unit": "MSamples/s (wide)", "channel_samples_per_s_M": round(8 * len(c3) / dt / 1e6, 3), "cores": 1, "kind": "reference",
                     "seconds": round(dt, 3), "sample": "flac -8 -e -p on the 10 s 8-channel 24-bit/96 kHz programme (raw input in /dev/shm), one pass"}
        # the clip workload's draw (same generator and seed as config_c5), its first clips
        rng = np.random.default_rng(777)
        nall = 10000
        dur = rng.uniform(1.0, 30.0, nall)
        bps = np.where(rng.random(nall) < 0.5, 16, 24)
        rate = np.where(rng.random(nall) < 0.5, 44100, 48000)
        ch = np.where(rng.random(nall) < 0.5, 1, 2)
        progs = {}
        wide = 0
        cmds = []
        for i in range(clips):
            key = (int(ch[i]), int(bps[i]), int(rate[i]))
            if key not in progs:
                c, b, r = key
                progs[key] = synth.music_like(30.0, rate=r, seed=5)[:, :c] if b == 16 else synth.multichannel_24(30.0, rate=r, channels=c, seed=9)
            c, b, r = key
            ln = int(dur[i] * r)
            name = "clip%d.raw" % i
            with open(os.path.join(tmp, name), "wb") as f:
                f.write(np.ascontiguousarray(progs[key][:ln].astype("<i2")).tobytes() if b == 16 else _raw24(progs[key][:ln]))
            cmds.append([exe, "-s", "-f", "-5", "--force-raw-format", "--endian=little", "--sign=signed", "--channels=%d" % c, "--bps=%d" % b, "--sample-rate=%d" % r,
                         "-o", "clip%d.flac" % i, name])
            wide += ln
        t0 = time.perf_counter()
        for cmd in cmds:
            r_ = subprocess.run(cmd, cwd=tmp, capture_output=True, text=True)
            if r_.returncode != 0:
                raise RuntimeError("%s: %s" % (" ".join(cmd), r_.stderr[-200:]))
        dt = time.perf_counter() - t0
        out["c5_clips"] = {"value": round(wide / dt / 1e6, 3), "unit": "MSamples/s (wide)", "clips_per_s": round(clips / dt, 2), "cores": 1, "kind": "reference", "seconds": round(dt, 3),
                           "sample": "flac -5, one process per clip, on the first %d clips of the workload's draw (raw files in /dev/shm), wall clock incl. process start" % clips}
    except Exception as ex:   # what was measured stands
        out["error"] = "%s: %s" % (type(ex).__name__, ex)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return out


def api_rates(base_pcm, minutes, ndev=1):
    """Rates through the boundary a libFLAC client sees (tools/api_rate.cpp, a compiled C++ client of the
    C ABI run as a child process): PCM in pageable host memory as int32, frames to a C write callback.
    PCIe-inclusive, so never `value`."""
    import subprocess
    import tempfile
    exe = os.path.join(ROOT, "tools", "bin", "api_rate")
    if not os.path.exists(exe):
        return {"error": "tools/bin/api_rate not built"}
    with tempfile.NamedTemporaryFile(suffix=".raw", dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as f:
        base_pcm.astype(np.int16).tofile(f)
        f.flush()
        try:
            out = subprocess.run([exe, f.name, str(minutes), "8", "2048", str(ndev)], capture_output=True, text=True, timeout=300)
        except subprocess.TimeoutExpired:
            return {"error": "api_rate timed out"}
    if out.returncode != 0:
        return {"error": "api_rate rc=%d: %s" % (out.returncode, out.stderr[-300:])}
    try:
        return json.loads(out.stdout.strip().splitlines()[-1])
    except ValueError:
        return {"error": "api_rate output not understood: %s" % out.stdout[-200:]}


# ---- the other BASELINE configurations, at their stated sizes (outside the timed region of the headline) ----------

def _timed(fn, reps):
    import torch
    fn()
    torch.cuda.synchronize()
    best = 1e30
    for _ in range(reps):
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best, out


def config_c3(dev, local, seconds=600):
    """flac -8 -e -p on 24-bit / 96 kHz / 8 channels: `seconds` of audio (a seeded 10 s programme tiled), PCM resident as int32."""
    import torch
    from flac_amd import engine, synth
    base = synth.multichannel_24(10.0, rate=96000, channels=8, seed=4321)
    reps = max(1, seconds // 10)
    n = len(base) * reps
    F = (n + FRAME - 1) // FRAME
    p = engine.make_params(8, 24, 96000, level=8, do_exhaustive_model_search=1, do_qlp_coeff_prec_search=1)
    enc = engine.Encoder(p, max_frames=F, device=local)
    enc.set_timing(True)
    d = torch.from_numpy(base).to(dev).repeat(reps, 1)
    cap = n * 8 * 4 + F * 256 + 4096       # the streams compress; a batch that does not says what it needs
    out = torch.empty(cap, dtype=torch.uint8, device=dev)
    fb = torch.empty(F, dtype=torch.int32, device=dev)
    dt, total = _timed(lambda: enc.encode_device(d.data_ptr(), engine.PCM_S32, n, out.data_ptr(), cap, fb.data_ptr()), 2)
    st = enc.last_timing()
    dom = max(st, key=st.get)
    algo = n * 8 * 3 + total
    res = {"workload": "flac -8 -e -p, 8 ch 24-bit/96 kHz, %d s (10 s programme tiled %d times), one GPU" % (n // 96000, reps),
           "value": round(n / dt / 1e6, 2), "unit": "MSamples/s (wide)", "channel_samples_per_s_M": round(8 * n / dt / 1e6, 1), "ms": round(dt * 1e3, 2),
           "frames": F, "compressed_bytes": total, "ratio": round(total / (n * 24), 4), "stage_ms": {k: round(v, 3) for k, v in st.items()},
           "algorithmic_bytes": algo, "dominant_stage": dom,
           "roofline": {"bound": "hbm", "achieved": round(algo / (st[dom] * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(algo / (st[dom] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)}}
    enc.close()
    return res


def config_c4(dev, local, base_pcm, hours=5):
    """Frame-parallel decode of a multi-GB -8 stream: `hours` of 16/44.1 stereo coded here, then decoded in one pass;
    also the rate at shorter batches of the same stream (the decode kernel is latency-bound: it should grow with the batch)."""
    import torch
    from flac_amd import engine
    base = torch.from_numpy(base_pcm.astype(np.int16)).to(dev)
    d_pcm = base.repeat(60 * hours, 1)
    n = d_pcm.shape[0]
    F = (n + FRAME - 1) // FRAME
    enc = engine.Encoder(engine.make_params(2, 16, RATE, level=8), max_frames=F, device=local)
    cap = n * 4 + F * 64 + 4096
    d_out = torch.empty(cap + 64, dtype=torch.uint8, device=dev)
    d_fb = torch.empty(F, dtype=torch.int32, device=dev)
    total = enc.encode_device(d_pcm.data_ptr(), engine.PCM_S16, n, d_out.data_ptr(), cap, d_fb.data_ptr())
    enc.close()
    sizes = d_fb.to(torch.int64)
    ends = torch.cumsum(sizes, 0)
    dec = engine.Decoder(2, 16, FRAME, int(F * 1.25) + 64, total + 64, device=local)
    d_back = torch.empty((n, 2), dtype=torch.int16, device=dev)
    res = {"workload": "decode of a %.2f GB flac -8 stream (%d h of 16/44.1 stereo, %d frames) resident in HBM, one pass, one GPU" % (total / 1e9, hours, F), "by_batch": {}}
    for h in sorted({1, 2, 4, 8, hours}):
        if h > hours:
            continue
        fr = min(F, (h * 3600 * RATE + FRAME - 1) // FRAME)
        nb = int(ends[fr - 1].item())
        ns = min(n, fr * FRAME)
        dt, info = _timed(lambda: dec.decode_device(d_out.data_ptr(), nb, d_back.data_ptr(), engine.PCM_S16, n), 3)
        ok = bool(torch.equal(d_back[:ns], d_pcm[:ns])) and info["frames"] == fr
        res["by_batch"]["%dh" % h] = {"value": round(ns / dt / 1e6, 1), "ms": round(dt * 1e3, 2), "bytes": nb, "bit_exact": ok}
        if h == hours:
            algo = nb + ns * 4
            res.update({"value": round(ns / dt / 1e6, 1), "unit": "MSamples/s", "ms": round(dt * 1e3, 2), "stream_bytes": nb, "frames": fr, "bit_exact": ok,
                        "algorithmic_bytes": algo,
                        "roofline": {"bound": "hbm", "achieved": round(algo / dt / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(algo / dt / 1e9 / HBM_PEAK_GBS, 5),
                                     "note": "whole decode pass (scan, parse, restore, CRC, chain, interleave) against bytes in + PCM out"}})
    dec.close()
    return res


def config_c5(dev, local, nclips=10000):
    """Batch encode of `nclips` clips of 1-30 s at -5, mixed 16/24-bit, 44.1/48 kHz, mono/stereo (seed 777): every clip a complete
    stream (head, MD5, frames) assembled on the device, one engine call per format group, the groups side by side; clips resident in HBM."""
    import torch
    from flac_amd import clips, engine, synth
    rng = np.random.default_rng(777)
    dur = rng.uniform(1.0, 30.0, nclips)
    bps = np.where(rng.random(nclips) < 0.5, 16, 24)
    rate = np.where(rng.random(nclips) < 0.5, 44100, 48000)
    ch = np.where(rng.random(nclips) < 0.5, 1, 2)
    plans = []
    audio_s = 0.0
    wide = 0
    for key in sorted({(int(c), int(b), int(r)) for c, b, r in zip(ch, bps, rate)}):
        c, b, r = key
        idx = np.nonzero((ch == c) & (bps == b) & (rate == r))[0]
        # four seeded 30 s programmes per format; a clip is the head of one of them
        if b == 16:
            progs = [torch.from_numpy(np.ascontiguousarray(synth.music_like(30.0, rate=r, seed=5 + k)[:, :c].astype(np.int16))).to(dev) for k in range(4)]
            fmt = engine.PCM_S16
        else:
            progs = [torch.from_numpy(np.ascontiguousarray(synth.multichannel_24(30.0, rate=r, channels=c, seed=9 + k))).to(dev) for k in range(4)]
            fmt = engine.PCM_S32
        lens = (dur[idx] * r).astype(np.int64)
        starts, total, nfr = clips.layout(lens, FRAME)
        d_pcm = torch.zeros((total, c), dtype=progs[0].dtype, device=dev)
        for j, (st, ln) in enumerate(zip(starts, lens)):
            d_pcm[int(st):int(st) + int(ln)] = progs[j % 4][:int(ln)]
        Fg = int(nfr.sum())
        enc = engine.Encoder(engine.make_params(c, b, r, level=5), max_frames=Fg, device=local)
        plans.append((enc, d_pcm, fmt, starts, lens, Fg))
        audio_s += float((lens / r).sum())
        wide += int(lens.sum())
    torch.cuda.synchronize()

    def run():
        return clips.encode_groups_device(plans, local)

    dt, outs = _timed(run, 2)
    total_bytes = int(sum(int(o[1][-1]) for o in outs))
    pcm_bytes = int(sum(p[1].numel() * (2 if p[2] == engine.PCM_S16 else 3) for p in plans))
    # spot check: the first clip of the first group decodes to what went in (the GPU tests compare whole batches with the reference)
    enc, d_pcm, fmt, starts, lens, Fg = plans[0]
    d_out, offs = outs[0]
    head = bytes(d_out[:4].cpu().numpy())
    for p in plans:
        p[0].close()
    return {"workload": "%d clips of 1-30 s at -5, mixed 16/24-bit, 44.1/48 kHz, mono/stereo (seed 777), %d format groups, clips resident in HBM" % (nclips, len(plans)),
            "value": round(wide / dt / 1e6, 1), "unit": "MSamples/s (wide)", "clips_per_s": round(nclips / dt, 1), "ms": round(dt * 1e3, 1), "audio_hours": round(audio_s / 3600, 2),
            "frames": int(sum(p[5] for p in plans)), "stream_bytes": total_bytes, "first_stream_magic_ok": head == b"fLaC",
            "algorithmic_bytes": pcm_bytes + total_bytes,
            "roofline": {"bound": "hbm", "achieved": round((pcm_bytes + total_bytes) / dt / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round((pcm_bytes + total_bytes) / dt / 1e9 / HBM_PEAK_GBS, 5), "note": "whole batch incl. per-clip MD5 on the device (one lane per clip) and stream assembly"}}


def sharded_stream(base, params, rank, world, local, dev, dist, host_group, rehearsal, hours=8):
    """Strong scaling of ONE long stream: `hours` of the headline programme cut into `world` frame ranges, rank r codes
    frames [r F/W, (r+1) F/W) (frames are independent given their samples and number: no collective on the data path), then
    the frames are brought together on rank 0 and laid end to end, by both routes north_star names:
      host  every rank downloads its frames to page-locked memory, the bytes go to rank 0 over the host (gloo) and are
            concatenated there - what the in-process pipe does with its lanes;
      rccl  sizes all-gathered and frame bytes gathered device to device over xGMI (RCCL), concatenated on rank 0's GPU.
    Both must give the same bytes for every W (length and CRC-32 reported)."""
    import zlib
    import torch
    from flac_amd import engine, sharding
    L = base.shape[0]
    nS = hours * 3600 * RATE
    F = (nS + FRAME - 1) // FRAME
    f0, f1 = sharding.frame_ranges(F, world, 1)[rank]
    s0, s1 = f0 * FRAME, min(f1 * FRAME, nS)
    ns = s1 - s0
    base_t = torch.from_numpy(np.ascontiguousarray(base.astype(np.int16))).to(dev)
    d_pcm = torch.empty((ns, 2), dtype=torch.int16, device=dev)   # this rank's range of the tiled programme
    at, src = 0, s0 % L
    while at < ns:
        k = min(L - src, ns - at)
        d_pcm[at:at + k] = base_t[src:src + k]
        at += k
        src = 0
    nf = f1 - f0
    enc = engine.Encoder(params, max_frames=max(nf, 1), device=local)
    cap = enc.max_output_bytes(max(nf, 1))
    d_out = torch.empty(cap, dtype=torch.uint8, device=dev)
    d_fb = torch.empty(max(nf, 1), dtype=torch.int32, device=dev)
    h_out = torch.empty(cap, dtype=torch.uint8).pin_memory()
    coll_dev = "cpu" if rehearsal else dev

    def encode():
        return enc.encode_device(d_pcm.data_ptr(), engine.PCM_S16, ns, d_out.data_ptr(), cap, d_fb.data_ptr(), final=(f1 == F), first_frame=f0) if nf else 0

    def counts_of(total, group, where):
        if world == 1:
            return [total]
        mine = torch.tensor([total], dtype=torch.int64, device=where)
        every = [torch.zeros(1, dtype=torch.int64, device=where) for _ in range(world)]
        dist.all_gather(every, mine, group=group)
        return [int(c[0]) for c in every]

    def host_route():
        total = encode()
        h_out[:total].copy_(d_out[:total], non_blocking=True)
        torch.cuda.synchronize()
        counts = counts_of(total, host_group, "cpu")
        if world == 1:
            return h_out[:total], total
        if rank == 0:
            whole = torch.empty(sum(counts), dtype=torch.uint8)
            whole[:counts[0]] = h_out[:counts[0]]
            at, reqs = counts[0], []
            for r in range(1, world):
                reqs.append(dist.irecv(whole[at:at + counts[r]], src=r, group=host_group))
                at += counts[r]
            for q in reqs:
                q.wait()
            return whole, int(whole.numel())
        dist.send(h_out[:total], dst=0, group=host_group)
        return None, 0

    def rccl_route():
        total = encode()
        counts = counts_of(total, None, coll_dev)
        if world == 1:
            return d_out[:total], total
        max_b = max(counts)
        piece = d_out[:max_b] if max_b <= cap else torch.cat([d_out, torch.zeros(max_b - cap, dtype=torch.uint8, device=dev)])
        piece = piece.to(coll_dev)
        got = [torch.empty(max_b, dtype=torch.uint8, device=coll_dev) for _ in range(world)] if rank == 0 else None
        dist.gather(piece, got, dst=0)
        if rank != 0:
            return None, 0
        whole = torch.cat([got[r][:counts[r]] for r in range(world)])
        return whole, int(whole.numel())

    def timed(route):
        route()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        whole, nbytes = route()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, whole, nbytes

    res = {"workload": "one stream of %d h of 16/44.1 stereo (%d frames) cut into %d frame ranges, one per GPU; frames brought together on rank 0 and laid end to end" % (hours, F, world),
           "scaling": "strong", "frames_per_rank": nf, "kernel_ms_per_rank_estimate": round(nf / 38760.0 * 4.0, 2)}
    for name, route in (("host", host_route), ("rccl", rccl_route)):
        dt, whole, nbytes = timed(route)
        if rank == 0:
            crc = zlib.crc32(memoryview(whole.cpu().numpy())) & 0xffffffff
            res[name] = {"value": round(nS / dt / 1e6, 1), "unit": "MSamples/s", "ms": round(dt * 1e3, 3), "stitched_bytes": nbytes, "stitched_crc32": "%08x" % crc}
        del whole
    enc.close()
    if rank != 0:
        return None
    res["host"]["includes"] = "encode on every rank + download to page-locked memory + bytes to rank 0 over the host (gloo send/recv of exact sizes) + concatenation on the host"
    res["rccl"]["includes"] = "encode on every rank + all_gather of sizes + gather of padded frame bytes to rank 0 over %s + concatenation on rank 0's GPU" % ("gloo (rehearsal on one GPU)" if rehearsal else "RCCL")
    res["same_bytes"] = res["host"]["stitched_bytes"] == res["rccl"]["stitched_bytes"] and res["host"]["stitched_crc32"] == res["rccl"]["stitched_crc32"]
    res["faster"] = "host" if res["host"]["ms"] <= res["rccl"]["ms"] else "rccl"
    return res


_WORK = {}


def _cpu_worker(kind, reps):
    import oracle_bindings as ob
    pcm32 = _WORK["pcm32"]
    if kind == "reference":
        _, rp = ob.make_params(2, 16, RATE, level=8)
        for _ in range(reps):
            ob.ref_encode(rp, pcm32)
    else:
        fp, _ = ob.make_params(2, 16, RATE, level=8)
        for _ in range(reps):
            ob.oracle_encode(fp, pcm32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--minutes", type=int, default=60, help="audio minutes per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-decode", action="store_true")
    ap.add_argument("--no-api", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the other BASELINE configurations (C3, C4, C5)")
    ap.add_argument("--clips", type=int, default=10000)
    ap.add_argument("--c3-seconds", type=int, default=600)
    ap.add_argument("--c4-hours", type=int, default=8)
    ap.add_argument("--no-sharded", action="store_true", help="skip the strong-scaling leg (one long stream cut into frame ranges)")
    ap.add_argument("--sharded-hours", type=int, default=8)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from flac_amd import engine

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    base, pcm = build_workload(60, args.minutes)
    nsamples = pcm.shape[0]
    nframes = (nsamples + FRAME - 1) // FRAME
    _WORK["pcm32"] = np.ascontiguousarray(base.astype(np.int32))
    # CPU baseline first: its worker processes are forked before this process touches the GPU
    cpu = None
    if not args.no_cpu_baseline and rank == 0:
        cpu = cpu_baseline(base)
        cpu["tool"] = cpu_tool_baselines(base)

    # API-level rates, in a child process of its own, before this process touches the GPU
    api = None
    if not args.no_api and rank == 0:
        api = api_rates(base, args.minutes)

    # rehearsal of the N > 1 path on a box with one GPU: all ranks on device 0, rendezvous and reductions over gloo
    rehearsal = os.environ.get("FLACAMD_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local = 0
    host_group = None
    if world > 1:
        import datetime
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        wait = datetime.timedelta(minutes=30)   # rank 0 measures the CPU baseline and the API rates before it joins
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world, timeout=wait)
            host_group = dist.group.WORLD
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local), timeout=wait)
            host_group = dist.new_group(backend="gloo", timeout=wait)   # the host route of the sharded leg
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU path")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)


    params = engine.make_params(2, 16, RATE, level=8)
    enc = engine.Encoder(params, max_frames=nframes, device=local)
    enc.set_timing(True)
    d_pcm = torch.from_numpy(pcm).to(dev)
    cap = enc.max_output_bytes(nframes)
    d_out = torch.empty(cap, dtype=torch.uint8, device=dev)
    d_fb = torch.empty(nframes, dtype=torch.int32, device=dev)
    first_frame = rank * nframes  # this rank's frame range of the N-hour stream

    def step():
        return enc.encode_device(d_pcm.data_ptr(), engine.PCM_S16, nsamples, d_out.data_ptr(), cap, d_fb.data_ptr(),
                                 final=True, first_frame=first_frame)

    total_bytes = 0
    for _ in range(args.warmup):
        total_bytes = step()

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # the timed steps are queued (flacamd_encode_device_queue): the host does not wait for a step before it launches the
    # next, as a caller coding batch after batch would not; every step is a full pass that writes all its frames, the
    # status and size of each are read by the wait call inside the timed region
    stage_ms = {}
    sync()
    t0 = time.perf_counter()
    done = 0
    while done < args.steps:
        group = min(QUEUE_DEPTH, args.steps - done)
        for _ in range(group):
            enc.encode_device_queue(d_pcm.data_ptr(), engine.PCM_S16, nsamples, d_out.data_ptr(), cap, d_fb.data_ptr(),
                                    final=True, first_frame=first_frame)
        totals = enc.wait()
        assert len(totals) == group and all(t == totals[0] for t in totals) and total_bytes in (0, totals[0]), (totals, total_bytes)
        total_bytes = totals[0]
        for k, v in enc.last_timing().items():   # mean over the group's batches
            stage_ms[k] = stage_ms.get(k, 0.0) + v * group
        done += group
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearsal else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- extra, outside the timed region: decode the stream just produced on the device and
    # check the round trip; compressed size against the reference's for the 60 s base programme
    decode = None
    if rank == 0 and not args.no_decode:
        try:
            dec = engine.Decoder(2, 16, FRAME, int(nframes * 1.25) + 64, total_bytes + 64, device=local)
            d_back = torch.empty((nsamples, 2), dtype=torch.int16, device=dev)
            d_in = d_out[:total_bytes]
            dec.decode_device(d_in.data_ptr(), total_bytes, d_back.data_ptr(), engine.PCM_S16, nsamples)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(3):
                info = dec.decode_device(d_in.data_ptr(), total_bytes, d_back.data_ptr(), engine.PCM_S16, nsamples)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t1) / 3
            decode = {"value": round(nsamples / dt / 1e6, 2), "unit": "MSamples/s", "ms": round(dt * 1e3, 3), "frames": info["frames"],
                      "round_trip_bit_exact": bool(torch.equal(d_back, d_pcm))}
            dec.close()
            del d_back
        except Exception as ex:  # reported, never hidden
            decode = {"error": str(ex)}
    del d_out, d_fb
    enc.close()
    del d_pcm
    torch.cuda.empty_cache()
    sharded = None
    if not args.no_sharded:
        try:
            sharded = sharded_stream(base, params, rank, world, local, dev, dist, host_group, rehearsal, args.sharded_hours)
        except Exception as ex:
            sharded = {"error": "%s: %s" % (type(ex).__name__, ex)}
        torch.cuda.empty_cache()
    # the in-process route a libFLAC client takes on a node: ONE encoder object over all N GPUs (flacamd_pipe lanes),
    # measured by rank 0's child process while the other ranks wait at the barrier below
    pipe = None
    if world > 1 and rank == 0 and not args.no_api:
        pipe = api_rates(base, args.minutes, ndev=1 if rehearsal else world)
    if world > 1:
        dist.barrier()
    configs = None
    if rank == 0 and world == 1 and not args.no_configs:
        configs = {}
        for name, fn in (("c3", lambda: config_c3(dev, local, args.c3_seconds)), ("c4_decode", lambda: config_c4(dev, local, base, args.c4_hours)),
                         ("c5_clips", lambda: config_c5(dev, local, args.clips))):
            try:
                torch.cuda.empty_cache()
                configs[name] = fn()
            except Exception as ex:  # reported, never hidden
                configs[name] = {"error": "%s: %s" % (type(ex).__name__, ex)}
    size_delta = None
    if rank == 0 and cpu is not None and "ref_bytes_60s" in cpu:
        enc60 = engine.Encoder(params, max_frames=(len(base) + FRAME - 1) // FRAME, device=local)
        frames60, _ = enc60.encode_host(base)
        enc60.close()
        size_delta = (sum(map(len, frames60)) - cpu["ref_bytes_60s"]) / cpu["ref_bytes_60s"]

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = world * nsamples / (elapsed / args.steps) / 1e6
        stage_ms = {k: v / args.steps for k, v in stage_ms.items()}
        dom = max(stage_ms, key=stage_ms.get)
        # algorithmic bytes of one launch of any stage = native-width PCM in + FLAC bytes out for
        # the whole batch (every stage sweeps the whole batch once)
        algo_bytes = nsamples * 4 + total_bytes
        achieved = algo_bytes / (stage_ms[dom] * 1e-3) / 1e9
        traffic, traffic_source = pmc_traffic(KERNEL_OF_STAGE.get(dom, dom), args.minutes, stage_ms[dom])
        line = {
            "metric": "encode MSamples/s at -8, 16/44.1 stereo",
            "value": round(value, 2), "unit": "MSamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int32", "data": "synthetic",
            "config": {"workload": "flac -8 (LPC order 12, blocksize 4096, mid/side, tukey(0.5)) on %d min of 16/44.1 stereo per GPU (%d different seeded 60 s synthetic programmes end to end, tiled to the length), PCM resident in HBM as int16; "
                                   "kernel-level figure: the stream's MD5 (one serial chain, computed on a host thread by the encoder object) and the transfers over PCIe are excluded - "
                                   "the same hour through the libFLAC-style API with host buffers is `api_value` / `api`" % (args.minutes, min(DISTINCT_MINUTES, args.minutes)),
                       "frames_per_gpu": nframes, "wide_samples_per_gpu": nsamples, "parallelism": "frame ranges x%d" % world,
                       "launch": "steps queued back to back on one stream (flacamd_encode_device_queue), sizes and status read by flacamd_encoder_wait in groups of up to %d" % QUEUE_DEPTH,
                       "compressed_bytes_per_gpu": total_bytes, "ratio": round(total_bytes / (nsamples * 4), 4)},
            "stage_ms": {k: round(v, 4) for k, v in stage_ms.items()},
            "roofline": {"bound": "hbm", "limited_by": "integer VALU issue (see note)", "kernel": KERNEL_OF_STAGE.get(dom, dom), "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": traffic_source,
                         "algorithmic_bytes_per_launch": algo_bytes,
                         "note": "integer VALU-bound kernel: ~(order+5) lane-ops per sample per LPC candidate, so the HBM fraction is small by nature; stage_ms has the per-stage times"},
        }
        if decode is not None:
            line["decode"] = decode
        if size_delta is not None:
            line["size_delta_vs_ref"] = round(size_delta, 6)
        if sharded is not None:
            line["sharded"] = sharded
        tool = (cpu or {}).get("tool") or {}
        if configs is not None:
            for name in ("c3", "c4_decode", "c5_clips"):   # the reference's number beside each GPU number
                if name in configs and name in tool:
                    configs[name]["cpu_baseline"] = tool[name]
            line["configs"] = configs
        if api is not None:
            line["api"] = api
            if isinstance(api.get("stream_md5_on"), (int, float)):
                # the headline workload as a client of the unchanged API sees it with the reference's defaults (MD5 on)
                line["api_value"] = api["stream_md5_on"]
                line["api_value_note"] = "FLAC__stream_encoder_process_interleaved from pageable host memory to a write callback, MD5 on (the reference's default), one GPU; MSamples/s"
        if pipe is not None:
            line["pipe"] = dict(pipe, note="one encoder object over %d GPU(s) from one process (flacamd_stream_encoder_set_devices / flacamd_pipe lanes), host buffers in, write callback out" % pipe.get("devices", 0))
        if cpu is not None:
            line["cpu_baseline"] = cpu
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
