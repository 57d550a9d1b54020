import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
from flac_amd import synth, engine
import oracle_bindings as ob

def run(name, pcm, ch, bps, rate, **kw):
    fp, rp = ob.make_params(ch, bps, rate, **kw)
    kw2 = dict(kw)
    p = engine.make_params(ch, bps, rate, **kw2)
    o = ob.oracle_encode(fp, pcm, want_info=True)
    bs = fp.blocksize or (4096 if fp.max_lpc_order else 1152)
    F = (len(pcm) + bs - 1)//bs
    enc = engine.Encoder(p, max_frames=F)
    enc.set_timing(True)
    t = time.time()
    frames, last = enc.encode_host(pcm)
    dt = time.time() - t
    same = sum(a == b for a, b in zip(frames, o['frames']))
    print(('OK ' if same == len(o['frames']) == len(frames) else 'BAD'), name, 'frames', len(frames), 'identical', same, 'bytes', sum(map(len, frames)), sum(map(len, o['frames'])), '%.3fs' % dt, enc.last_timing())
    if same != len(frames):
        # white-box: compare autoc / fixed_err / winners
        G = F * (4 if (ch == 2 and fp.do_mid_side) else ch)
        lag = fp.max_lpc_order + 1
        nsig = G // F
        if fp.max_lpc_order:
            au = enc.debug(0, np.float32, fp.num_apodizations * lag * G).reshape(fp.num_apodizations, lag, G)
        fe = enc.debug(1, np.uint64, 5 * G).reshape(5, G)
        wh = enc.debug(2, np.uint32, 4 * G).reshape(G, 4)
        bad = 0
        for f in range(F):
            info = o['infos'][f]
            if frames[f] == o['frames'][f]:
                continue
            for s in range(min(nsig, 4)):
                g = f * nsig + s
                oa = np.array(info.autoc[s][:lag], dtype=np.float32)
                ofe = np.array(info.fixed_err[s][:], dtype=np.uint64)
                msg = []
                if fp.max_lpc_order and not np.array_equal(oa.view(np.uint32), au[0, :, g].view(np.uint32)):
                    msg.append('autoc differs %s vs %s' % (oa[:3], au[0, :3, g]))
                if not np.array_equal(ofe, fe[:, g]):
                    msg.append('fixed_err differs %s vs %s' % (ofe, fe[:, g]))
                h = wh[g]
                msg.append('gpu win type %d order %d prec %d shift %d po %d rice2 %d wasted %d est %d bps %d cand %d' % (h[0] & 3, (h[0] >> 2) & 63, (h[0] >> 8) & 31, (h[0] >> 13) & 31, (h[0] >> 18) & 15, (h[0] >> 22) & 1, (h[0] >> 23) & 31, h[1], h[2], h[3]))
                print('  frame', f, 'sig', s, '; '.join(msg))
            print('  oracle: assignment', info.channel_assignment, [(info.sub[c].type, info.sub[c].order, info.sub[c].qlp_precision, info.sub[c].qlp_shift, info.sub[c].partition_order, info.sub[c].estimate_bits, info.sub[c].wasted_bits) for c in range(ch)], 'len', len(frames[f]), len(o['frames'][f]))
            bad += 1
            if bad >= 3:
                break
    enc.close()

print('devices', engine.device_count())
m = synth.music_like(6.0, seed=1234)
run('st16 -8', m, 2, 16, 44100, level=8)
run('st16 -8 s16 input', m.astype(np.int16), 2, 16, 44100, level=8)
for lvl in range(8):
    run('st16 -%d' % lvl, m, 2, 16, 44100, level=lvl)
run('len 4097', m[:4097], 2, 16, 44100, level=8)
run('len 5000', m[:5000], 2, 16, 44100, level=8)
run('mono', m[:, :1], 1, 16, 44100, level=8)
pcm24 = synth.multichannel_24(0.3)
run('24/96 8ch -8', pcm24, 8, 24, 96000, level=8)
run('24/96 8ch -8 -e -p', pcm24[:12000], 8, 24, 96000, level=8, do_qlp_coeff_prec_search=1)
