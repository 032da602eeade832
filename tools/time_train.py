"""GPU box: bench.py's training leg alone (batch 4096, 200 steps without a host sync) -- one line per run, for A/B of kernel settings
passed through the environment."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from mural_amd.data import PackedGenome  # noqa: E402

dev = torch.device("cuda", 0)
codes = bench.synthetic_genome(4_096_000 + 2000)
packed, mask = bench.pack2(codes)
genome = PackedGenome(packed, mask, len(codes), dev)
out = bench.train_steps_per_s(dev, genome)
tag = " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("MURAL_"))
print("%s: %.1f steps/s  %.3f ms/step (synchronised %.3f ms)" % (tag or "default", out["steps_per_s"], out["ms_per_step"], out["ms_per_step_synchronised"]))
