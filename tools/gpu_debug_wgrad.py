"""Diagnostic: the MFMA weight-gradient kernel against torch's conv gradients (float64) over the U-Net's shapes, and its timing against
the vector-ALU kernel (MURAL_WGRAD_MFMA=0 in a second process)."""
import os
os.environ.setdefault("MURAL_HIP_FLAVOR", "debug")      # validation hooks / development switches: the debug flavour of the library
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mural_amd.model.indel_train import Conv

cases = [(3, 4, 8, 7, 1, 3, 1, 500), (2, 8, 16, 7, 4, 3, 1, 501), (2, 32, 40, 7, 5, 3, 1, 77), (5, 40, 48, 7, 2, 3, 1, 4),
         (2, 48, 40, 7, 1, 3, 2, 9), (2, 16, 8, 7, 1, 3, 4, 130), (3, 48, 96, 5, 1, 2, 1, 70), (3, 96, 48, 1, 1, 0, 1, 70),
         (1, 4, 4, 7, 1, 3, 1, 2000), (4, 8, 16, 5, 1, 2, 1, 8000), (4, 16, 8, 1, 1, 0, 1, 8000), (3, 24, 48, 5, 1, 2, 1, 400),
         (3, 40, 80, 5, 1, 2, 1, 80), (7, 80, 40, 1, 1, 0, 1, 16), (7, 48, 96, 5, 1, 2, 1, 4), (2, 8, 8, 1, 1, 0, 1, 8000),
         (5, 16, 32, 5, 1, 2, 1, 2000), (3, 4, 4, 7, 1, 3, 1, 8000), (2, 32, 16, 3, 1, 1, 1, 1040), (9, 8, 20, 5, 1, 2, 1, 48)]
rng = torch.Generator().manual_seed(5)
worst = 0.0
for B, Cin, Cout, K, stride, pad, up, L in cases:
    x = torch.randn((B, Cin, L), generator=rng)
    w = torch.randn((Cout, Cin, K), generator=rng) / (Cin * K) ** 0.5
    b = torch.randn(Cout, generator=rng)
    xr, wr, br = x.double().requires_grad_(), w.double().requires_grad_(), b.double().requires_grad_()
    xu = xr.repeat_interleave(up, dim=2) if up > 1 else xr
    yr = torch.nn.functional.conv1d(xu, wr, br, stride=stride, padding=pad)
    g = torch.randn(yr.shape, generator=rng)
    yr.backward(g.double())
    xd, wd, bd = x.cuda().requires_grad_(), w.cuda().requires_grad_(), b.cuda().requires_grad_()
    yd = Conv.apply(xd, wd, bd, stride, pad, up)
    yd.backward(g.cuda())
    rel = lambda got, want: float((got.cpu().double() - want).abs().max() / max(1.0, float(want.abs().max())))
    ew, eb = rel(wd.grad, wr.grad), rel(bd.grad, br.grad)
    worst = max(worst, ew if ew == ew else 1.0, eb if eb == eb else 1.0)
    print((B, Cin, Cout, K, stride, pad, up, L), "dW %.2e db %.2e" % (ew, eb), flush=True)
print("worst %.2e" % worst)

if worst > 2e-6:
    sys.exit(1)
if os.environ.get("TIME"):
    from mural_amd.model import train_ops as T
    shapes = [(128, 8, 16, 5, 1, 2, 1, 8000), (128, 16, 8, 1, 1, 0, 1, 8000), (128, 4, 8, 7, 1, 3, 1, 8000), (128, 16, 32, 5, 1, 2, 1, 2000),
              (128, 24, 48, 5, 1, 2, 1, 400), (128, 48, 24, 1, 1, 0, 1, 400), (128, 32, 64, 5, 1, 2, 1, 80), (128, 48, 96, 5, 1, 2, 1, 4),
              (128, 8, 16, 7, 4, 3, 1, 8000), (128, 16, 8, 7, 1, 3, 4, 2000)]
    for B, Cin, Cout, K, stride, pad, up, L in shapes:
        x = torch.randn((B, Cin, L), device="cuda")
        w = torch.randn((Cout, Cin, K), device="cuda")
        Lout = (L * up + 2 * pad - K) // stride + 1
        dy = torch.randn((B, Cout, Lout), device="cuda")
        dW = torch.empty_like(w)
        db = torch.empty(Cout, device="cuda")
        from mural_amd import _lib
        part = torch.empty(int(_lib.lib().mural_op_convg_bwd_scratch(Cin, Cout, K)), device="cuda")
        run = lambda: T._call("mural_op_convg_bwd", dy, x, w, B, Cin, L, Cout, K, stride, pad, up, None, dW, db, part, part.numel(), T._stream(x))
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(20):
            run()
        torch.cuda.synchronize()
        print("time", (B, Cin, Cout, K, stride, pad, up, L), "%.1f us" % ((time.perf_counter() - t) / 20 * 1e6), flush=True)
