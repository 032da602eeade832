"""GPU box: the config-5 legs of bench.py alone (file-to-file e2e at the bench's size with the rank_share measurement)."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

if __name__ == "__main__":
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    out = bench.config5_e2e(dev)
    print(json.dumps(out))
