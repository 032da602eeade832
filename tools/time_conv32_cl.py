"""GPU box: time single launches of the channel-last conv kernels (csrc/conv32_cl.hip; WHICH=cw: csrc/conv32_wave.hip) at the
training shapes."""
import ctypes as C
import os
os.environ.setdefault("MURAL_HIP_FLAVOR", "debug")      # validation hooks / development switches: the debug flavour of the library
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mural_amd import _lib  # noqa: E402

lib = _lib.lib()
dev = torch.device("cuda", 0)
B = 4096
CW = os.environ.get("WHICH", "cl") == "cw"
tag = ("cw " if CW else "cl ") + " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("MURAL_"))
for L in (134, 67, 20, 23, 7):
    x = torch.randn(B, L, 32, device=dev)
    dy = torch.randn(B, L, 32, device=dev)
    r1, r2 = torch.randn(B, L, 32, device=dev), torch.randn(B, L, 32, device=dev)
    y = torch.empty_like(x)
    W = torch.randn(32, 32, 3, device=dev) * 0.1
    bias, gamma, beta = torch.randn(32, device=dev), torch.rand(32, device=dev) + 0.5, torch.randn(32, device=dev)
    state = torch.empty(4, 32, device=dev)
    rm, rv = torch.zeros(32, device=dev), torch.ones(32, device=dev)
    acc = torch.zeros(32 * 2 * 32, dtype=torch.float64, device=dev)
    acc_out = torch.zeros_like(acc)
    part = torch.empty(1024 * 3104, device=dev)
    nrow = C.c_int32(0)
    st = _lib.current_stream_ptr(dev)
    wfs_t = torch.empty(6144, device=dev)
    wfs = wfs_t.data_ptr() if os.environ.get("WFRAG", "1") == "1" else None
    if CW:
        _lib.check(lib.mural_debug_cw_wfrag(W.data_ptr(), wfs_t.data_ptr(), st))
    _lib.check(lib.mural_debug_cl_bn_stats(x.data_ptr(), B * L, 1, acc.data_ptr(), st))

    def fwd(a, b):
        if CW:
            return lambda: _lib.check(lib.mural_debug_cw_conv32_fwd(x.data_ptr(), B, L, 1, acc.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                                                    rm.data_ptr(), rv.data_ptr(), state.data_ptr(), W.data_ptr(), bias.data_ptr(), 0, a, b,
                                                                    acc_out.data_ptr(), 1, y.data_ptr(), wfs, st))
        return lambda: _lib.check(lib.mural_debug_cl_conv32_fwd(x.data_ptr(), B, L, 1, acc.data_ptr(), gamma.data_ptr(), beta.data_ptr(), rm.data_ptr(),
                                                                rv.data_ptr(), state.data_ptr(), W.data_ptr(), bias.data_ptr(), 0, a, b,
                                                                acc_out.data_ptr(), 1, y.data_ptr(), st))

    def bwd():
        if CW:
            _lib.check(lib.mural_debug_cw_conv32_bwd(dy.data_ptr(), x.data_ptr(), W.data_ptr(), B, L, state.data_ptr(), gamma.data_ptr(), 1, y.data_ptr(),
                                                     acc_out.data_ptr(), part.data_ptr(), C.byref(nrow), wfs, st))
        else:
            _lib.check(lib.mural_debug_cl_conv32_bwd(dy.data_ptr(), x.data_ptr(), W.data_ptr(), B, L, state.data_ptr(), 1, y.data_ptr(),
                                                     acc_out.data_ptr(), part.data_ptr(), C.byref(nrow), st))

    res = []
    for name, fn in (("fwd", fwd(None, None)), ("fwd+r1", fwd(r1.data_ptr(), None)), ("fwd+r1+r2", fwd(r1.data_ptr(), r2.data_ptr())), ("bwd", bwd)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            fn()
        torch.cuda.synchronize()
        res.append("%s %.1f" % (name, (time.perf_counter() - t0) / 50 * 1e6))
    print(f"[{tag}] L={L:4d} ({B * 32 * L * 4 / 1e6:.0f} MB/tensor) us: " + "  ".join(res))
