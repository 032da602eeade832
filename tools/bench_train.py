"""GPU-box measurement of the SNV training step (BASELINE.json configs[2]): S-config from scratch, batch 4096,
synthetic labelled sites, Adam lr 1e-3, CE-sum, clip 10, dropouts at defaults.  Prints steps/s and sites/s."""
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
from mural_amd.train import clip_grad_norm_  # noqa: E402


def main(B=4096, steps=10, warmup=3):
    dev = torch.device("cuda", 0)
    codes = bench.synthetic_genome(4_096_000 + 2000)
    packed, mask = bench.pack2(codes)
    genome = PackedGenome(packed, mask, len(codes), dev)
    model = bench.build_model(dev).train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, fused=True)       # one multi-tensor launch; same update rule
    crit = nn.CrossEntropyLoss(reduction="sum")
    rng = np.random.default_rng(1)
    labels = torch.from_numpy(rng.choice(4, size=(steps + warmup) * B, p=[0.955, 0.015, 0.015, 0.015])).to(dev)
    cont = torch.zeros(B, 1, device=dev)
    times = []
    for s in range(steps + warmup):
        idx = torch.arange(s * B, (s + 1) * B, device=dev)
        pos, strand = idx + 1000, (idx & 1).to(torch.uint8)
        cat = genome.encode_kmer(pos, strand, 10, 3)
        x = genome.encode_onehot(pos, strand, 1000)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        preds = model((cont, cat), x)
        loss = crit(preds, labels[s * B:(s + 1) * B])
        opt.zero_grad()
        loss.backward()
        clip_grad_norm_(model, 10)
        opt.step()
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    t = float(np.median(times[warmup:]))
    print("train step B=%d: %.1f ms/step = %.2f steps/s = %.0f sites/s; loss %.1f; %.1f TFLOP/s algorithmic (22.6 MFLOP/site)"
          % (B, t * 1e3, 1 / t, B / t, loss.item(), B / t * 22.6e6 / 1e12))


if __name__ == "__main__":
    main()
