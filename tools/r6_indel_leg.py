"""GPU box: bench.py's INDEL forward leg alone (packed + dense), twice."""
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
for _ in range(2):
    r = bench.indel_positions_per_s(dev, genome)
    print(json.dumps({k: r[k] for k in ("positions_per_s", "positions_per_s_dense_input") if k in r}))
