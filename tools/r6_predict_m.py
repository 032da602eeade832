"""GPU box: model_predict_m over a HOST loader of 16-row batches (bench.py variants.model_predict_m_batch16_loader) and the 16-site
dense calls, alone."""
import os
import sys
import time

import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from mural_amd.model import model_predict_m  # noqa: E402

dev = torch.device("cuda", 0)
model = bench.build_model(dev)
g = torch.Generator(device=dev).manual_seed(5)
B = 16384
codes = torch.randint(0, 4, (B, 2001), device=dev, generator=g)
x = torch.nn.functional.one_hot(codes, 4).permute(0, 2, 1).float().contiguous()
c = codes[:, 990:1011]
cat = (c[:, :-2] * 16 + c[:, 1:-1] * 4 + c[:, 2:]).contiguous()
cont = torch.zeros(B, 1, device=dev, dtype=torch.float64)
ys = torch.zeros(B, 1)
loader = [(ys[i:i + 16], cont[i:i + 16].cpu(), cat[i:i + 16].cpu(), x[i:i + 16].cpu()) for i in range(0, B, 16)]
crit = nn.CrossEntropyLoss(reduction="sum")
with torch.no_grad():
    model_predict_m(model, loader[:64], crit, dev, 4)
    for rep in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        model_predict_m(model, loader, crit, dev, 4)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("model_predict_m host loader: %.0f rows/s (%.1f ms)" % (B / dt, dt * 1e3))
    calls = [(cont[i * 16:(i + 1) * 16], cat[i * 16:(i + 1) * 16].contiguous(), x[i * 16:(i + 1) * 16].contiguous()) for i in range(256)]
    for _ in range(50):
        model((calls[0][0], calls[0][1]), calls[0][2])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(2000):
        co, ca, xx = calls[i % 256]
        model((co, ca), xx)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 2000
    print("16-site dense calls: %.1f us per call = %.0f bases/s" % (dt * 1e6, 16 / dt))
    for nsite in (1, 16, 64, 256):
        co, ca, xx = cont[:nsite], cat[:nsite].contiguous(), x[:nsite].contiguous()
        for _ in range(50):
            model((co, ca), xx)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(1000):
            model((co, ca), xx)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 1000
        print("%d-site dense calls: %.1f us per call = %.0f bases/s" % (nsite, dt * 1e6, nsite / dt))
