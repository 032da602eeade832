"""GPU box: the channel-last conv kernels of the training step (csrc/conv32_cl.hip; WHICH=cw: the wave-private kernels of
csrc/conv32_wave.hip) against torch ops in float64 on the CPU:
forward with BatchNorm finalisation, residuals and fused batch sums; backward with weight / bias gradient, input gradient and the
BatchNorm-backward sums.  Prints the worst relative error per quantity; exits non-zero above 1e-4."""
import ctypes as C
import os
os.environ.setdefault("MURAL_HIP_FLAVOR", "debug")      # validation hooks / development switches: the debug flavour of the library
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mural_amd import _lib  # noqa: E402

lib = _lib.lib()
dev = torch.device("cuda", 0)
worst = 0.0
CW = os.environ.get("WHICH", "cl") == "cw"


def rel(a, b):
    return float((a.double().cpu() - b).abs().max() / (b.abs().max() + 1e-12))


def acc_sum(acc):
    return acc.view(32, 2, 32).sum(0).cpu()


CASES = ((12, 20, 1), (12, 7, 0), (12, 134, 1), (32, 23, 0), (5, 67, 1), (64, 8, 1), (1100, 20, 1), (3, 1, 0), (700, 134, 1))
if CW:
    # (the last four: 4 / 9 / 3 / 2 rows per unit = 6 / 5 / 7 / 8 blocks)
    CASES += ((4096, 20, 1), (4099, 23, 1), (2051, 7, 0), (1500, 67, 1), (3000, 134, 1), (9, 142, 1), (2500, 33, 1),
              (9000, 20, 1), (20000, 7, 0), (6500, 33, 1), (5000, 60, 1))
for case_no, (B, L, PRE) in enumerate(CASES):
    g = torch.Generator().manual_seed(B * 1000 + L)
    x = torch.randn(B, L, 32, generator=g, dtype=torch.float64)
    W = torch.randn(32, 32, 3, generator=g, dtype=torch.float64) * 0.2
    bias = torch.randn(32, generator=g, dtype=torch.float64)
    gamma = torch.rand(32, generator=g, dtype=torch.float64) + 0.5
    beta = torch.randn(32, generator=g, dtype=torch.float64)
    r1 = torch.randn(B, L, 32, generator=g, dtype=torch.float64)
    r2 = torch.randn(B, L, 32, generator=g, dtype=torch.float64)
    dy = torch.randn(B, L, 32, generator=g, dtype=torch.float64)
    # ---- reference (NCL inside torch)
    xn = x.permute(0, 2, 1)
    ax = F.relu(xn) if PRE else xn
    mean = ax.mean(dim=(0, 2))
    var = ax.var(dim=(0, 2), unbiased=False)
    invstd = 1.0 / torch.sqrt(var + 1e-5)
    xhat = (ax - mean[None, :, None]) * invstd[None, :, None]
    a = xhat * gamma[None, :, None] + beta[None, :, None]
    a.requires_grad_(True)
    Wp = W.clone().requires_grad_(True)
    bp = bias.clone().requires_grad_(True)
    y_ref = F.conv1d(a, Wp, bp, padding=1) + r1.permute(0, 2, 1) + r2.permute(0, 2, 1)
    (y_ref * dy.permute(0, 2, 1)).sum().backward()
    dz_ref = a.grad
    # ---- device
    f32 = lambda t: t.to(torch.float32).contiguous().to(dev)
    xd, Wd, bd, gd, betad, r1d, r2d, dyd = map(f32, (x, W, bias, gamma, beta, r1, r2, dy))
    st = _lib.current_stream_ptr(dev)
    wfs = torch.empty(6144, device=dev)      # odd cases: fragments from the per-step relayout kernel, even: gathered in LDS
    if CW:
        _lib.check(lib.mural_debug_cw_wfrag(Wd.data_ptr(), wfs.data_ptr(), st))
    acc = torch.zeros(32 * 2 * 32, dtype=torch.float64, device=dev)
    _lib.check(lib.mural_debug_cl_bn_stats(xd.data_ptr(), B * L, PRE, acc.data_ptr(), st))
    s = acc_sum(acc)
    e = max(rel(s[0], ax.sum(dim=(0, 2))), rel(s[1], (ax * ax).sum(dim=(0, 2))))
    state = torch.empty(4, 32, device=dev)
    rm, rv = torch.zeros(32, device=dev), torch.ones(32, device=dev)
    acc_out = torch.zeros(32 * 2 * 32, dtype=torch.float64, device=dev)
    yd = torch.empty_like(xd)
    if CW:
        _lib.check(lib.mural_debug_cw_conv32_fwd(xd.data_ptr(), B, L, PRE, acc.data_ptr(), gd.data_ptr(), betad.data_ptr(), rm.data_ptr(), rv.data_ptr(),
                                                 state.data_ptr(), Wd.data_ptr(), bd.data_ptr(), 0, r1d.data_ptr(), r2d.data_ptr(), acc_out.data_ptr(),
                                                 1, yd.data_ptr(), wfs.data_ptr() if case_no % 2 else None, st))
    else:
        _lib.check(lib.mural_debug_cl_conv32_fwd(xd.data_ptr(), B, L, PRE, acc.data_ptr(), gd.data_ptr(), betad.data_ptr(), rm.data_ptr(), rv.data_ptr(),
                                                 state.data_ptr(), Wd.data_ptr(), bd.data_ptr(), 0, r1d.data_ptr(), r2d.data_ptr(), acc_out.data_ptr(),
                                                 1, yd.data_ptr(), st))
    yr = y_ref.detach().permute(0, 2, 1)
    if os.environ.get("MURAL_TEST_VERBOSE"):          # round-off of the forward next to torch's own float32 path on the CPU
        xf = x.float().permute(0, 2, 1)
        af = F.batch_norm(F.relu(xf) if PRE else xf, None, None, gamma.float(), beta.float(), True, 0.1, 1e-5)
        y32 = F.conv1d(af, W.float(), bias.float(), padding=1) + r1.float().permute(0, 2, 1) + r2.float().permute(0, 2, 1)
        print("   forward mean |err|: hip %.2e  torch32 %.2e" % (float((yd.double().cpu() - yr).abs().mean()),
                                                               float((y32.double().permute(0, 2, 1) - yr).abs().mean())))
    so = acc_sum(acc_out)
    e_f = max(rel(yd, yr), rel(so[0], F.relu(yr).sum(dim=(0, 1))), rel(so[1], (F.relu(yr) ** 2).sum(dim=(0, 1))),
              rel(state[2], mean), rel(state[3], invstd))
    dzd = torch.empty_like(xd)
    stat = torch.zeros(32 * 2 * 32, dtype=torch.float64, device=dev)
    part = torch.empty(1024 * 3104, device=dev)
    nrow = C.c_int32(0)
    if CW:
        _lib.check(lib.mural_debug_cw_conv32_bwd(dyd.data_ptr(), xd.data_ptr(), Wd.data_ptr(), B, L, state.data_ptr(), gd.data_ptr(), PRE,
                                                 dzd.data_ptr(), stat.data_ptr(), part.data_ptr(), C.byref(nrow),
                                                 wfs.data_ptr() if case_no % 2 else None, st))
    else:
        _lib.check(lib.mural_debug_cl_conv32_bwd(dyd.data_ptr(), xd.data_ptr(), Wd.data_ptr(), B, L, state.data_ptr(), PRE, dzd.data_ptr(),
                                                 stat.data_ptr(), part.data_ptr(), C.byref(nrow), st))
    pr = part[:nrow.value * 3104].view(nrow.value, 3104).double().sum(0).cpu()
    ss = acc_sum(stat)
    dzr = dz_ref.permute(0, 2, 1)
    errs = {"stats": e, "fwd": e_f, "dz": rel(dzd, dzr), "dW": rel(pr[:3072].view(32, 32, 3), Wp.grad), "db": rel(pr[3072:], bp.grad),
            "sum dz": rel(ss[0], dz_ref.sum(dim=(0, 2))), "sum dz*xhat": rel(ss[1], (dz_ref * xhat).sum(dim=(0, 2)))}
    worst = max(worst, max(errs.values()))
    print(f"B={B:5d} L={L:4d} relu={PRE} " + "  ".join(f"{k} {v:.1e}" for k, v in errs.items()))
print("worst %.2e" % worst)
sys.exit(0 if worst <= 1e-4 else 1)
