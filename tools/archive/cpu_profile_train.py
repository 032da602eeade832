"""Host-side profile (cProfile) of the SNV training step: where the Python / launch overhead goes."""
import cProfile
import os
import pstats
import sys

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from mural_amd.data import PackedGenome  # noqa: E402

dev = torch.device("cuda", 0)
B = 4096
codes = bench.synthetic_genome(200_000 + 2000)
packed, mask = bench.pack2(codes)
genome = PackedGenome(packed, mask, len(codes), dev)
model = bench.build_model(dev).train()
opt = torch.optim.Adam(model.parameters(), lr=1e-3, fused=True)
crit = nn.CrossEntropyLoss(reduction="sum")
labels = torch.zeros(B, dtype=torch.int64, device=dev)
cont = torch.zeros(B, 1, device=dev)
idx = torch.arange(B, device=dev)
pos, strand = idx + 1000, (idx & 1).to(torch.uint8)
cat = genome.encode_kmer(pos, strand, 10, 3)
x = genome.encode_onehot(pos, strand, 1000)


def step():
    loss = crit(model((cont, cat), x), labels)
    opt.zero_grad()
    loss.backward()
    torch.nn.utils.clip_grad_norm_(model.parameters(), 10)
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(30):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
