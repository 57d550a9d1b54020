import subprocess, sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flac_amd import synth
base = synth.music_like(60.0, seed=1).astype(np.int16)
base.tofile("/dev/shm/base_probe.raw")
exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bin", "api_rate")
for rep in (1, 2, 3, 4, 8):
    for batch in ("2048", "512"):
        out = subprocess.run([exe, "/dev/shm/base_probe.raw", str(rep), "8", batch, "1"], capture_output=True, text=True, timeout=200)
        print(rep, batch, out.returncode, out.stderr.strip()[-120:], out.stdout[:60])
os.remove("/dev/shm/base_probe.raw")
