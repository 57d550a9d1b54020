import glob
import json
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden_names():
    return sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))


def load_golden(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)
    meta = json.loads(bytes(z["meta"]).decode())
    sizes = [int(s) for s in z["sizes"]]
    blob = bytes(z["frames"])
    frames, off = [], 0
    for s in sizes:
        frames.append(blob[off:off + s])
        off += s
    return dict(meta=meta, pcm=z["pcm"].astype(np.int32), frames=frames, md5=bytes(z["md5"]))
