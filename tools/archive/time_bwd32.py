"""GPU box: time one backward launch of the 32->32 conv layer (mural_op_conv32_bwd) and one forward launch at the training shapes."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mural_amd import _lib  # noqa: E402

lib = _lib.lib()
dev = torch.device("cuda", 0)
B = 4096
for L in (134, 67, 20):
    dy = torch.randn(B, 32, L, device=dev)
    x = torch.randn(B, 32, L, device=dev)
    W = torch.randn(32, 32, 3, device=dev) * 0.1
    st = torch.rand(4, 32, device=dev) + 0.5
    dW, db, dz = torch.empty_like(W), torch.empty(32, device=dev), torch.empty_like(x)
    acc = torch.zeros(32, 2, 32, dtype=torch.float64, device=dev)
    part = torch.empty(int(lib.mural_op_conv32_wgrad_scratch()), device=dev)
    stream = _lib.current_stream_ptr(dev)

    def bwd():
        _lib.check(lib.mural_op_conv32_bwd(dy.data_ptr(), x.data_ptr(), W.data_ptr(), B, L, st[0].data_ptr(), st[1].data_ptr(), 1,
                                           st[2].data_ptr(), st[3].data_ptr(), dW.data_ptr(), db.data_ptr(), dz.data_ptr(),
                                           acc.data_ptr(), part.data_ptr(), part.numel(), stream))

    def fwd():
        _lib.check(lib.mural_op_conv32(x.data_ptr(), W.data_ptr(), db.data_ptr(), dz.data_ptr(), B, L, 0, st[0].data_ptr(), st[1].data_ptr(),
                                       1, 0, dy.data_ptr(), None, 1, 1, None, None, None, acc.data_ptr(), stream))

    for name, fn in (("bwd", bwd), ("fwd+res+stats", fwd)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 50
        print(f"L={L:4d} {name:14s} {dt * 1e6:8.1f} us   tensor {B * 32 * L * 4 / 1e6:.0f} MB")
