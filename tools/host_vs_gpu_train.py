"""Training step: host enqueue time vs completion time (is the step launch-bound or GPU-bound?)."""
import os
import sys
import time

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


for _ in range(5):
    step()
torch.cuda.synchronize()
N = 30
t0 = time.perf_counter()
for _ in range(N):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("per step: host enqueue %.2f ms, until the GPU is done %.2f ms (no sync inside the loop; the forward's deferred input "
      "check waits for the encode kernel only)" % ((t1 - t0) / N * 1e3, (t2 - t0) / N * 1e3))
