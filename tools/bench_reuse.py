"""GPU box: dense same-strand variant -- reuse path vs per-window path on the bench workload."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from mural_amd.data import PackedGenome  # noqa: E402

dev = torch.device("cuda", 0)
codes = bench.synthetic_genome(bench.GENOME_SITES + 2000)
packed, mask = bench.pack2(codes)
genome = PackedGenome(packed, mask, len(codes), dev)
model = bench.build_model(dev)
B = int(os.environ.get("B", 1_000_000))
steps = int(os.environ.get("STEPS", 5))
idx = torch.arange(B, device=dev, dtype=torch.int64)
pos, strand = idx + 1000, (idx & 1).to(torch.uint8)
which = os.environ.get("WHICH", "both")
for name, fn in (("per-window", model.forward_packed), ("reuse", model.forward_packed_reuse)):
    if which not in ("both", name):
        continue
    with torch.no_grad():
        fn(genome, pos, strand, 10, 3)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            out = fn(genome, pos, strand, 10, 3)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
    print(f"{name:12s} {B / dt / 1e6:8.2f} M sites/s   {dt * 1e3:8.2f} ms per {B} sites", flush=True)
