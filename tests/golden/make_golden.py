#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the REAL reference (oracle/_ref/libflacref.so, compiled
from /root/reference by oracle/Makefile).  Run in the build container only:

    python tests/golden/make_golden.py

Each fixture holds the input PCM, the encoder settings and the reference's output: every
frame's bytes, the STREAMINFO MD5 and min/max frame size.  They are data, not code."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle_bindings as ob  # noqa: E402
from cases import parity_cases  # noqa: E402

KEEP = ["st16-level0", "st16-level1", "st16-level3", "st16-level5", "st16-level8", "st16-8-p", "st16-8-bs576", "st16-8-bs17",
        "st16-8-l32", "st16-8-r8", "mono16-8", "wasted", "white", "fsd3", "sine8", "sine20", "sine24-st", "silence", "multi-apod",
        "disable-all", "rate12345", "loose-4", "8ch24-8", "8ch24-8-e-p", "len5", "len4097"]


def main():
    assert ob.have_ref(), "build oracle/_ref first (make -C oracle ref)"
    for name, make, ch, bps, rate, kw in parity_cases(small=True):
        if name not in KEEP:
            continue
        pcm = make()
        if name.startswith("loose"):
            pcm = pcm[:60000]
        elif len(pcm) > 3 * 4096 + 100 and not name.startswith("st16-8-bs"):
            pcm = pcm[:3 * 4096 + 100]
        fp, rp = ob.make_params(ch, bps, rate, **kw)
        r = ob.ref_encode(rp, pcm)
        sizes = np.array([len(f) for f in r["frames"]], dtype=np.uint32)
        store = pcm.astype(np.int16) if bps <= 16 else pcm.astype(np.int32)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), pcm=store, frames=np.frombuffer(b"".join(r["frames"]), dtype=np.uint8),
                            sizes=sizes, md5=np.frombuffer(r["md5"], dtype=np.uint8),
                            meta=np.frombuffer(json.dumps(dict(name=name, channels=ch, bps=bps, rate=rate, params={k: v for k, v in kw.items()},
                                                                min_framesize=r["min_framesize"], max_framesize=r["max_framesize"])).encode(), dtype=np.uint8))
        print(name, pcm.shape, len(r["frames"]), int(sizes.sum()))


if __name__ == "__main__":
    main()
