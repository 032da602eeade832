"""GPU box: host enqueue time of the training step against its device time (bench.py's composition, symbol windows)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from mural_amd.data import PackedGenome  # noqa: E402
from mural_amd.train import Adam, CrossEntropySum, clip_grad_norm_  # noqa: E402

dev = torch.device("cuda", 0)
codes = bench.synthetic_genome(4_096_000 + 2000)
packed, mask = bench.pack2(codes)
genome = PackedGenome(packed, mask, len(codes), dev)
B = 4096
model = bench.build_model(dev).train()
opt = Adam(model.parameters(), lr=1e-3)
crit = CrossEntropySum()
labels = torch.zeros(B, dtype=torch.int64, device=dev)
cont = torch.zeros(B, 1, device=dev)
idx = torch.arange(B, device=dev)
pos, strand = idx + 1000, (idx & 1).to(torch.uint8)


def step():
    cat = genome.encode_kmer(pos, strand, 10, 3)
    x = genome.encode_symbols(pos, strand, 1000)
    loss = crit(model((cont, cat), x), labels)
    opt.zero_grad()
    loss.backward()
    clip_grad_norm_(model, 10)
    opt.step()


for _ in range(30):
    step()
torch.cuda.synchronize()
N = 300
t0 = time.perf_counter()
for _ in range(N):
    step()
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print("host enqueue %.3f ms/step, total %.3f ms/step (%.1f steps/s)" % (t_host / N * 1e3, t_all / N * 1e3, N / t_all))
# host alone: the same loop with the device kept idle is not possible; instead time each host part of one step, synchronised
parts = {}
for _ in range(50):
    torch.cuda.synchronize(); t = time.perf_counter()
    cat = genome.encode_kmer(pos, strand, 10, 3); x = genome.encode_symbols(pos, strand, 1000)
    parts["encode"] = parts.get("encode", 0) + time.perf_counter() - t; torch.cuda.synchronize(); t = time.perf_counter()
    out = model((cont, cat), x)
    parts["forward"] = parts.get("forward", 0) + time.perf_counter() - t; torch.cuda.synchronize(); t = time.perf_counter()
    loss = crit(out, labels)
    parts["loss"] = parts.get("loss", 0) + time.perf_counter() - t; torch.cuda.synchronize(); t = time.perf_counter()
    opt.zero_grad(); loss.backward()
    parts["backward"] = parts.get("backward", 0) + time.perf_counter() - t; torch.cuda.synchronize(); t = time.perf_counter()
    clip_grad_norm_(model, 10)
    parts["clip"] = parts.get("clip", 0) + time.perf_counter() - t; torch.cuda.synchronize(); t = time.perf_counter()
    opt.step()
    parts["adam"] = parts.get("adam", 0) + time.perf_counter() - t
print("host time per part, us (device idle at each start):", {k: round(v / 50 * 1e6) for k, v in parts.items()})
