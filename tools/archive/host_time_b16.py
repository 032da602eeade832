"""GPU box: host time of the C entry point of the 16-site dense call (mural_snv_forward_dense) and of its Python wrapper."""
import ctypes as C
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from mural_amd import _lib  # noqa: E402

dev = torch.device("cuda", 0)
model = bench.build_model(dev)
B = 16
codes = torch.randint(0, 4, (B, 2001), device=dev)
x = torch.nn.functional.one_hot(codes, 4).permute(0, 2, 1).float().contiguous()
c = codes[:, 990:1011]
cat = (c[:, :-2] * 16 + c[:, 1:-1] * 4 + c[:, 2:]).contiguous()
cont = torch.zeros(B, 1, device=dev, dtype=torch.float64)
lib = _lib.lib()
with torch.no_grad():
    for _ in range(50):
        model((cont, cat), x)
    torch.cuda.synchronize()
    handle = model._get_handle()
    out = torch.empty((B, 4), device=dev)
    ws = model._workspace(B, dev)
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    st = _lib.current_stream_ptr(dev)
    n = 2000
    for which in ("C call", "wrapper"):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            if which == "C call":
                lib.mural_snv_forward_dense(handle, cat.data_ptr(), x.data_ptr(), B, out.data_ptr(), ws.data_ptr(), ws.numel(), status.data_ptr(), st)
            else:
                model((cont, cat), x)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"{which}: host {(t1 - t0) / n * 1e6:.1f} us/call, until done {(t2 - t0) / n * 1e6:.1f} us/call")
