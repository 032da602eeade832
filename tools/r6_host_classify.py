"""GPU box (host side only): mural_host_dense_to_symbols on 8192 windows of 2001 columns in 16-row batches vs thread count."""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mural_amd import _lib  # noqa: E402

L, n = 2001, 8192
codes = torch.randint(0, 4, (n, L))
big = torch.nn.functional.one_hot(codes, 4).permute(0, 2, 1).float().contiguous()
bs = [big[i:i + 16].clone() for i in range(0, n, 16)]
nb = len(bs)
ptrs = (C.c_void_p * nb)(*[b.data_ptr() for b in bs])
counts = (C.c_int64 * nb)(*[16] * nb)
out = torch.empty(n * L, dtype=torch.uint8).pin_memory()
bad = C.c_int64(0)
print("host cpus", os.cpu_count())
for th in ("8", "16", "32", "64", "128"):
    os.environ["MURAL_HOST_THREADS"] = th
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        _lib.check(_lib.lib().mural_host_dense_to_symbols(ptrs, counts, nb, L, out.data_ptr(), C.byref(bad)))
        ts.append(time.perf_counter() - t0)
    print(th, "threads: %.2f ms (min %.2f)" % (np.median(ts) * 1e3, min(ts) * 1e3))
t0 = time.perf_counter()
for _ in range(5):
    y = torch.cat([b[:, 0, :19] for b in bs])
print("torch.cat of 512 small tensors: %.2f ms" % ((time.perf_counter() - t0) / 5 * 1e3))
