"""GPU box (under rocprofv3 --kernel-trace): packed-entry long-window calls at R = 4000, N windows per call (argv[1], default 2048)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from mural_amd.data import PackedGenome  # noqa: E402

dev = torch.device("cuda", 0)
R, n = int(os.environ.get("LW_R", "4000")), int(sys.argv[1]) if len(sys.argv) > 1 else 2048
codes = bench.synthetic_genome(2_000_000 + 2 * R)
packed, mask = bench.pack2(codes)
genome = PackedGenome(packed, mask, len(codes), dev)
model = bench.build_model(dev, R)
idx = torch.arange(n, device=dev, dtype=torch.int64) * 401
pos, strand = idx + R, (idx & 1).to(torch.uint8)
with torch.no_grad():
    for _ in range(12):
        model.forward_packed(genome, pos, strand, local_radius=bench.LOCAL_RADIUS, local_order=bench.LOCAL_ORDER)
torch.cuda.synchronize()
