"""GPU box: host time spent inside the two C calls of the training step (launch cost of the composed kernels)."""
import os
import sys
import time

import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from mural_amd import _lib  # noqa: E402
from mural_amd.data import PackedGenome  # noqa: E402
from mural_amd.train import clip_grad_norm_  # noqa: E402

os.environ["MURAL_DEBUG_NO_INPUT_CHECK"] = "1"
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
lib = _lib.lib()
spent = {"mural_snv_train_forward": 0.0, "mural_snv_train_backward": 0.0}


class Timed:
    def __init__(self, inner):
        self.inner = inner

    def __getattr__(self, name):
        fn = getattr(self.inner, name)
        if name not in spent:
            return fn

        def wrapped(*a):
            t = time.perf_counter()
            r = fn(*a)
            spent[name] += time.perf_counter() - t
            return r
        return wrapped


timed = Timed(lib)
_lib.lib = lambda: timed
seg = {"fwd": 0.0, "loss": 0.0, "bwd": 0.0, "clip": 0.0, "opt": 0.0}


def step():
    t0 = time.perf_counter()
    out = model((cont, cat), x)
    t1 = time.perf_counter()
    loss = crit(out, labels)
    opt.zero_grad()
    t2 = time.perf_counter()
    loss.backward()
    t3 = time.perf_counter()
    clip_grad_norm_(model, 10)
    t4 = time.perf_counter()
    opt.step()
    t5 = time.perf_counter()
    for k, v in zip(seg, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)):
        seg[k] += v


for _ in range(5):
    step()
torch.cuda.synchronize()
for k in spent:
    spent[k] = 0.0
for k in seg:
    seg[k] = 0.0
N = 30
t0 = time.perf_counter()
for _ in range(N):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
print("host per step %.3f ms; segments (ms): %s" % ((t1 - t0) / N * 1e3, {k: round(v / N * 1e3, 3) for k, v in seg.items()}))
print("inside the C calls (ms):", {k: round(v / N * 1e3, 3) for k, v in spent.items()})
