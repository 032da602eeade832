"""GPU box: the long-window legs of bench.py's variants (R = 4000 dense / packed at 2048 and 512 windows, R = 8000, R = 16000)."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from mural_amd.data import PackedGenome  # noqa: E402

dev = torch.device("cuda", 0)
codes = bench.synthetic_genome(bench.GENOME_SITES + 2 * bench.DISTAL_RADIUS)
packed, mask = bench.pack2(codes)
genome = PackedGenome(packed, mask, len(codes), dev)
model = bench.build_model(dev)
out = bench.workload_variants(dev, model, genome)
print(json.dumps({k: {kk: vv for kk, vv in v.items() if kk in ("frac_of_fp32_peak", "bases_per_s", "ms_per_call")} for k, v in out.items() if "long" in k}))
