"""GPU box: the reference's default call pattern (16 sites per forward): host enqueue time vs end-to-end time per call."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

dev = torch.device("cuda", 0)
model = bench.build_model(dev)
B = 16
g = torch.Generator(device=dev).manual_seed(5)
codes = torch.randint(0, 4, (B * 64, 2001), device=dev, generator=g)
x = torch.nn.functional.one_hot(codes, 4).permute(0, 2, 1).float().contiguous()
c = codes[:, 990:1011]
cat = (c[:, :-2] * 16 + c[:, 1:-1] * 4 + c[:, 2:]).contiguous()
cont = torch.zeros(B * 64, 1, device=dev, dtype=torch.float64)
calls = [(cont[i * B:(i + 1) * B], cat[i * B:(i + 1) * B].contiguous(), x[i * B:(i + 1) * B].contiguous()) for i in range(64)]
with torch.no_grad():
    for i in range(100):
        co, ca, xx = calls[i % 64]
        model((co, ca), xx)
    if "--freeze" in sys.argv:
        from mural_amd._host import freeze_host_heap
        freeze_host_heap()
    torch.cuda.synchronize()
    n = 3000
    t0 = time.perf_counter()
    for i in range(n):
        co, ca, xx = calls[i % 64]
        model((co, ca), xx)
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f"host enqueue {t_host / n * 1e6:.1f} us/call, end to end {t_all / n * 1e6:.1f} us/call = {B * n / t_all / 1e3:.1f} k bases/s")
    t0 = time.perf_counter()
    for i in range(300):
        co, ca, xx = calls[i % 64]
        model((co, ca), xx)
        torch.cuda.synchronize()
    print(f"synchronised call latency {(time.perf_counter() - t0) / 300 * 1e6:.1f} us")
