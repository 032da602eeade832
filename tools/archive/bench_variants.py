"""GPU-box measurements of the secondary workload variants SURVEY.md 8(d) asks to be reported next to bench.py's line:

  * dense: "independent random windows" — B x 2001 i.i.d. bases per site handed over as the reference's own tensors
    (cat_x int64 (B,19), distal_x fp32 one-hot (B,4,2001)), no shared genome buffer;
  * b16:   the reference's default predict batch of 16 sites per forward call (commands/predict.py:90), packed input;
  * train: BASELINE.json configs[2] — S-config from scratch, batch 4096, Adam, dropouts at defaults, N timed steps;
  * indel: BASELINE.json configs[3] — UNet_Small human-insertion geometry (L=8000, 8 classes, use_reverse) from the packed genome.

Prints one JSON object per variant.  Usage: python tools/archive/bench_variants.py [dense] [b16] [train] [indel]"""
import json
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
from mural_amd.model import model_choice, weights_init  # noqa: E402

dev = torch.device("cuda", 0)


def timed(fn, reps, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def genome_of(n):
    codes = bench.synthetic_genome(n)
    packed, mask = bench.pack2(codes)
    return PackedGenome(packed, mask, len(codes), dev)


def dense(B=16384):
    model = bench.build_model(dev)
    g = torch.Generator(device=dev).manual_seed(5)
    codes = torch.randint(0, 4, (B, 2001), device=dev, generator=g)
    x = torch.nn.functional.one_hot(codes, 4).permute(0, 2, 1).float().contiguous()
    c = codes[:, 990:1011]
    cat = (c[:, :-2] * 16 + c[:, 1:-1] * 4 + c[:, 2:]).contiguous()        # order-3 indices of the +-10 window
    cont = torch.zeros(B, 1, device=dev, dtype=torch.float64)
    with torch.no_grad():
        dt = timed(lambda: model((cont, cat), x), 10)
    return {"variant": "independent random windows, dense reference tensors (cat_x int64, distal_x fp32 one-hot)", "batch": B,
            "bases_per_s": B / dt, "ms_per_call": dt * 1e3, "input_GB_per_s": B * (4 * 2001 * 4 + 19 * 8) / dt / 1e9}


def b16():
    model = bench.build_model(dev)
    genome = genome_of(1_000_000 + 2000)
    idx = torch.arange(16, device=dev, dtype=torch.int64)
    calls = [(idx + 1000 + 16 * i, ((idx + 16 * i) & 1).to(torch.uint8)) for i in range(256)]
    it = iter(range(10 ** 9))

    def one():
        pos, strand = calls[next(it) % 256]
        return model.forward_packed(genome, pos, strand, local_radius=10, local_order=3)

    with torch.no_grad():
        dt = timed(one, 500, warm=20)
    return {"variant": "reference default predict batch (16 sites per forward call), packed input", "batch": 16,
            "bases_per_s": 16 / dt, "us_per_call": dt * 1e6}


def train(steps=200, warmup=20, B=4096):
    genome = genome_of(4_096_000 + 2000)
    model = bench.build_model(dev).train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, fused=True)       # one multi-tensor launch; same update rule
    crit = nn.CrossEntropyLoss(reduction="sum")
    rng = np.random.default_rng(1)
    labels = torch.from_numpy(rng.choice(4, size=(steps + warmup) * B, p=[0.955, 0.015, 0.015, 0.015])).to(dev)
    cont = torch.zeros(B, 1, device=dev)
    losses = []
    t0 = None
    for s in range(steps + warmup):
        if s == warmup:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        idx = (torch.arange(s * B, (s + 1) * B, device=dev)) % 4_096_000
        pos, strand = idx + 1000, (idx & 1).to(torch.uint8)
        cat = genome.encode_kmer(pos, strand, 10, 3)
        x = genome.encode_onehot(pos, strand, 1000)
        loss = crit(model((cont, cat), x), labels[s * B:(s + 1) * B])
        opt.zero_grad()
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 10)
        opt.step()
        if s % 20 == 0:
            losses.append(round(loss.item() / B, 4))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    return {"variant": "train from scratch, batch 4096, Adam lr 1e-3, CE-sum, clip 10, dropouts 0.1/0.1/0.25; window encode "
            "from the packed genome inside the timed loop", "steps": steps, "steps_per_s": 1 / dt, "ms_per_step": dt * 1e3,
            "sites_per_s": B / dt, "algorithmic_TFLOPs": B / dt * 22.6e6 / 1e12, "loss_per_site_every_20_steps": losses}


def indel(n=8192):
    cfg = dict(CNN_out_channels=8, CNN_kernel_size=7, down_list=[1, 4, 5, 5, 5, 2], use_reverse=True)
    torch.manual_seed(0)
    model = model_choice(0, cfg, dict(n_class=8), "indel")
    model.apply(weights_init)
    model = model.to(dev).eval()
    genome = genome_of(1_000_000 + 8000)
    idx = torch.arange(n, device=dev, dtype=torch.int64)
    pos, strand = idx * 100 + 4000, (idx & 1).to(torch.uint8)
    with torch.no_grad():
        dt = timed(lambda: model.forward_packed(genome, pos, strand, 4000), 3, warm=1)
    return {"variant": "UNet_Small insertion geometry (L=8000, 8 classes, use_reverse), packed input", "positions": n,
            "positions_per_s": n / dt, "algorithmic_TFLOPs": n / dt * 113.4e6 / 1e12}


if __name__ == "__main__":
    which = sys.argv[1:] or ["dense", "b16", "train", "indel"]
    for w in which:
        print(json.dumps({w: globals()[w]()}), flush=True)
