"""GPU box: the tower-parallel small-call kernel (two workgroups per site meet through a device-scope counter) under repetition: every
call of 1 .. 256 sites must give the rows the big-batch path gives, 3000 calls back to back, no synchronisation in between."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from mural_amd.data import PackedGenome  # noqa: E402

dev = torch.device("cuda", 0)
codes = bench.synthetic_genome(400_000 + 2000)
packed, mask = bench.pack2(codes)
genome = PackedGenome(packed, mask, len(codes), dev)
model = bench.build_model(dev)
g = torch.Generator(device=dev).manual_seed(1)
allpos = torch.randint(1000, 400_000, (300_000,), device=dev, generator=g)
allstr = (allpos & 1).to(torch.uint8)
with torch.no_grad():
    ref = model.forward_packed(genome, allpos, allstr, 10, 3)
    worst, outs, o = 0.0, [], 0
    sizes = [1, 16, 64, 256, 3, 200, 17, 255]
    for it in range(3000):
        n = sizes[it % len(sizes)]
        outs.append((o, n, model.forward_packed(genome, allpos[o:o + n], allstr[o:o + n], 10, 3)))
        o = (o + n) % 299_000
    torch.cuda.synchronize()
    bad = 0
    for o, n, out in outs:
        d = float((out - ref[o:o + n]).abs().max())
        worst = max(worst, d)
        bad += int(not torch.isfinite(out).all()) + int(d > 1e-5)
print("calls", len(outs), "worst abs diff of log-probabilities", worst, "bad calls", bad)
