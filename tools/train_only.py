"""GPU box: bench.py's SNV training leg alone (free-running steps, the figure of `train.steps_per_s`), for A/B runs of development
switches (debug flavour): python tools/train_only.py [steps]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MURAL_HIP_FLAVOR", "debug")
import torch  # noqa: E402

import bench  # noqa: E402
from mural_amd.data import PackedGenome  # noqa: E402

dev = torch.device("cuda", 0)
codes = bench.synthetic_genome(4_096_000 + 2000)
packed, mask = bench.pack2(codes)
genome = PackedGenome(packed, mask, len(codes), dev)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
out = bench.train_steps_per_s(dev, genome, steps=steps, sync_steps=min(30, steps))
print(json.dumps({k: round(v, 2) for k, v in out.items() if k.startswith("steps_per_s")}))
