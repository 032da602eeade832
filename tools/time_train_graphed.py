"""GPU box: mural_amd.train.GraphedTrainStep at batch 4096 (whole step replayed as one HIP graph), inputs encoded per step."""
import os
import sys
import time

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from mural_amd.data import PackedGenome  # noqa: E402
from mural_amd.train import GraphedTrainStep  # noqa: E402

dev = torch.device("cuda", 0)
B = 4096
codes = bench.synthetic_genome(4_096_000 + 2000)
packed, mask = bench.pack2(codes)
genome = PackedGenome(packed, mask, len(codes), dev)
model = bench.build_model(dev).train()
opt = torch.optim.Adam(model.parameters(), lr=1e-3, capturable=True, fused=True)
crit = nn.CrossEntropyLoss(reduction="sum")
rng = np.random.default_rng(1)
steps, warmup = (int(sys.argv[1]) if len(sys.argv) > 1 else 200), 10
labels = torch.from_numpy(rng.choice(4, size=(steps + warmup) * B, p=[0.955, 0.015, 0.015, 0.015])).to(dev)
cont = torch.zeros(B, 1, device=dev)


def batch(s):
    idx = torch.arange(s * B, (s + 1) * B, device=dev)
    pos, strand = idx + 1000, (idx & 1).to(torch.uint8)
    return cont, genome.encode_kmer(pos, strand, 10, 3), genome.encode_onehot(pos, strand, 1000), labels[s * B:(s + 1) * B]


g = GraphedTrainStep(model, opt, crit, *batch(0))
for s in range(warmup):
    g(*batch(s))
torch.cuda.synchronize()
t0 = time.perf_counter()
for s in range(warmup, warmup + steps):
    loss = g(*batch(s))
g.finish()
t = (time.perf_counter() - t0) / steps
print("graphed train step B=%d: %.3f ms/step = %.1f steps/s; loss/site %.4f" % (B, t * 1e3, 1 / t, loss.item() / B))
