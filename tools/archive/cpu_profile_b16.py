"""GPU box: host-side profile (cProfile) of the 16-site dense predict call."""
import cProfile
import os
import pstats
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

dev = torch.device("cuda", 0)
model = bench.build_model(dev)
B = 16
codes = torch.randint(0, 4, (B, 2001), device=dev)
x = torch.nn.functional.one_hot(codes, 4).permute(0, 2, 1).float().contiguous()
c = codes[:, 990:1011]
cat = (c[:, :-2] * 16 + c[:, 1:-1] * 4 + c[:, 2:]).contiguous()
cont = torch.zeros(B, 1, device=dev, dtype=torch.float64)
with torch.no_grad():
    for _ in range(200):
        model((cont, cat), x)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(2000):
        model((cont, cat), x)
    pr.disable()
    torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
