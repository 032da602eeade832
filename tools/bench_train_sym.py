"""GPU box: the SNV training step exactly as bench.py's `train` leg composes it (symbol windows from the packed genome +
mural_amd.train.CrossEntropySum + mural_amd.train.clip_grad_norm_ + fused Adam), 13 steps at batch 4096 -- the command of the round-5
training profile (tools/archive/profile_r05.sh), so that hbm_bytes_per_step / launches_per_step belong to the loop whose rate is reported."""
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


def main(B=4096, steps=10, warmup=3):
    dev = torch.device("cuda", 0)
    codes = bench.synthetic_genome(4_096_000 + 2000)
    packed, mask = bench.pack2(codes)
    genome = PackedGenome(packed, mask, len(codes), dev)
    model = bench.build_model(dev).train()
    opt = Adam(model.parameters(), lr=1e-3)
    crit = CrossEntropySum()
    rng = np.random.default_rng(1)
    labels = torch.from_numpy(rng.choice(4, size=(steps + warmup) * B, p=[0.955, 0.015, 0.015, 0.015])).to(dev)
    cont = torch.zeros(B, 1, device=dev)
    idx_all = torch.arange((steps + warmup) * B, device=dev)
    pos_all, strand_all = idx_all + 1000, (idx_all & 1).to(torch.uint8)
    times = []
    for s in range(steps + warmup):
        pos, strand = pos_all[s * B:(s + 1) * B], strand_all[s * B:(s + 1) * B]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        cat = genome.encode_kmer(pos, strand, 10, 3)
        x = genome.encode_symbols(pos, strand, 1000)
        loss = crit(model((cont, cat), x), labels[s * B:(s + 1) * B])
        opt.zero_grad()
        loss.backward()
        clip_grad_norm_(model, 10)
        opt.step()
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    t = float(np.median(times[warmup:]))
    print("train step (symbol route) B=%d: %.2f ms/step = %.1f steps/s; loss %.1f" % (B, t * 1e3, 1 / t, loss.item()))


if __name__ == "__main__":
    main()
