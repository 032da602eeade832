"""The training step fed with dense one-hot windows (the reference loader's form) against symbol windows, alternating, with the
per-kernel picture of the dense route's extras.  usage: python tools/archive/r5_dense_route.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from mural_amd.data import PackedGenome
from mural_amd.train import CrossEntropySum, clip_grad_norm_
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device("cuda", 0)
codes = bench.synthetic_genome(bench.GENOME_SITES + 2 * bench.DISTAL_RADIUS + 1)
packed, mask = bench.pack2(codes)
genome = PackedGenome(packed, mask, len(codes), dev)
B = 4096
model = bench.build_model(dev).train()
opt = torch.optim.Adam(model.parameters(), lr=1e-3, fused=True)
crit = CrossEntropySum()
rng = np.random.default_rng(1)
labels = torch.from_numpy(rng.choice(4, size=B, p=[0.955, 0.015, 0.015, 0.015]).astype(np.int64)).to(dev)
cont = torch.zeros(B, 1, device=dev)
idx = torch.arange(B, device=dev)
pos, strand = idx + bench.DISTAL_RADIUS, (idx & 1).to(torch.uint8)


def step(dense):
    cat = genome.encode_kmer(pos, strand, bench.LOCAL_RADIUS, bench.LOCAL_ORDER)
    x = genome.encode_onehot(pos, strand, bench.DISTAL_RADIUS) if dense else genome.encode_symbols(pos, strand, bench.DISTAL_RADIUS)
    loss = crit(model((cont, cat), x), labels)
    opt.zero_grad()
    loss.backward()
    clip_grad_norm_(model, 10)
    opt.step()


for d in (False, True):
    for _ in range(5):
        step(d)
for rep in range(3):
    for d in (False, True):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step(d)
        torch.cuda.synchronize()
        print("%s: %.1f steps/s" % ("dense " if d else "symbol", steps / (time.perf_counter() - t0)))
