"""GPU check of the conv32 MFMA kernels (forward with pre-op / residuals / fused sums, input gradient, weight gradient)
against torch ops computed on the CPU in float64, over a few (B, L) shapes."""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mural_amd import _lib  # noqa: E402
from mural_amd.model import train_ops as T  # noqa: E402

dev = torch.device("cuda", 0)
torch.manual_seed(0)
worst = 0.0
for B, L in [(32, 20), (32, 7), (32, 134), (5, 67), (300, 23), (4096, 20), (2050, 134), (1, 8)]:
    x = torch.randn(B, 32, L)
    W = torch.randn(32, 32, 3) * 0.1
    bias = torch.randn(32)
    s, t = torch.rand(32) + 0.5, torch.randn(32)
    r1, r2 = torch.randn(B, 32, L), torch.randn(B, 32, L)
    xin = s.view(1, 32, 1) * F.relu(x) + t.view(1, 32, 1)
    ref = F.conv1d(xin.double(), W.double(), bias.double(), padding=1) + r1.double() + r2.double()
    xg, Wg, bg, sg, tg, r1g, r2g = [v.to(dev) for v in (x, W, bias, s, t, r1, r2)]
    y = torch.empty(B, 32, L, device=dev)
    acc = torch.zeros(32, 2, 32, dtype=torch.float64, device=dev)
    st = _lib.current_stream_ptr(dev)
    T._call("mural_op_conv32", xg, Wg, bg, y, B, L, 0, sg, tg, 1, 0, r1g, r2g, 1, 1, None, None, None, acc, st)
    e = (y.cpu().double() - ref).abs().max().item()
    a = acc.sum(0).cpu()
    rr = F.relu(ref)
    e1 = ((a[0] - rr.sum((0, 2))).abs().max() / (rr.sum((0, 2)).abs().max() + 1)).item()
    e2 = ((a[1] - (rr * rr).sum((0, 2))).abs().max() / ((rr * rr).sum((0, 2)).abs().max() + 1)).item()
    # input gradient + BN-backward sums
    dy = torch.randn(B, 32, L)
    mean, invstd = torch.randn(32) * 0.1, torch.rand(32) + 0.5
    dref = F.conv_transpose1d(dy.double(), W.double(), padding=1)
    xh = (F.relu(x.double()) - mean.double().view(1, 32, 1)) * invstd.double().view(1, 32, 1)
    dz = torch.empty(B, 32, L, device=dev)
    acc2 = torch.zeros(32, 2, 32, dtype=torch.float64, device=dev)
    T._call("mural_op_conv32", dy.to(dev), Wg, None, dz, B, L, 1, None, None, 0, 0, None, None, 2, 1, xg, mean.to(dev),
            invstd.to(dev), acc2, st)
    e3 = (dz.cpu().double() - dref).abs().max().item()
    a2 = acc2.sum(0).cpu()
    e4 = ((a2[0] - dref.sum((0, 2))).abs().max() / (dref.sum((0, 2)).abs().max() + 1)).item()
    e5 = ((a2[1] - (dref * xh).sum((0, 2))).abs().max() / ((dref * xh).sum((0, 2)).abs().max() + 1)).item()
    # weight gradient
    xin64 = xin.double().requires_grad_(False)
    Wd = W.double().requires_grad_(True)
    bd = bias.double().requires_grad_(True)
    out = F.conv1d(xin64, Wd, bd, padding=1)
    out.backward(dy.double())
    dW = torch.empty(32, 32, 3, device=dev)
    db = torch.empty(32, device=dev)
    part = torch.empty(int(_lib.lib().mural_op_conv32_wgrad_scratch()), device=dev)
    T._call("mural_op_conv32_wgrad", dy.to(dev), xg, B, L, sg, tg, 1, dW, db, part, part.numel(), st)
    e6 = ((dW.cpu().double() - Wd.grad).abs().max() / (Wd.grad.abs().max() + 1)).item()
    e7 = ((db.cpu().double() - bd.grad).abs().max() / (bd.grad.abs().max() + 1)).item()
    # fused backward (one pass over dy)
    dW2 = torch.empty(32, 32, 3, device=dev)
    db2 = torch.empty(32, device=dev)
    dz2 = torch.empty(B, 32, L, device=dev)
    acc3 = torch.zeros(32, 2, 32, dtype=torch.float64, device=dev)
    T._call("mural_op_conv32_bwd", dy.to(dev), xg, Wg, B, L, sg, tg, 1, mean.to(dev), invstd.to(dev), dW2, db2, dz2, acc3, part,
            part.numel(), st)
    a3 = acc3.sum(0).cpu()
    e8 = max(((dW2.cpu().double() - Wd.grad).abs().max() / (Wd.grad.abs().max() + 1)).item(),
             ((db2.cpu().double() - bd.grad).abs().max() / (bd.grad.abs().max() + 1)).item(),
             (dz2.cpu().double() - dref).abs().max().item(),
             ((a3[0] - dref.sum((0, 2))).abs().max() / (dref.sum((0, 2)).abs().max() + 1)).item(),
             ((a3[1] - (dref * xh).sum((0, 2))).abs().max() / ((dref * xh).sum((0, 2)).abs().max() + 1)).item())
    worst = max(worst, e8)
    print("   fused backward worst %.1e" % e8)
    print("B=%5d L=%3d  fwd %.1e  sum %.1e sq %.1e | dgrad %.1e s1 %.1e s2 %.1e | dW %.1e db %.1e" % (B, L, e, e1, e2, e3, e4, e5, e6, e7))
    worst = max(worst, e, e1, e2, e3, e4, e5, e6, e7)
print("worst", worst)
sys.exit(0 if worst < 1e-3 else 1)
