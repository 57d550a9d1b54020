"""Development aid: the 8-channel 24-bit -e -p configuration of bench.py alone (for profiling)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
print(json.dumps(bench.config_c3(dev, 0, int(sys.argv[1]) if len(sys.argv) > 1 else 120)))
