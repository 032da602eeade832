"""GPU debugging aid: run one training step of a golden fixture and check every mural_op_conv32 / conv32_wgrad call
against torch (CPU, float64) on the very tensors the call received."""
import os
import sys

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import _util as U  # noqa: E402
from tests.test_gpu_snv import product_from_hp  # noqa: E402
from mural_amd.model import train_ops as T  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "S"
fx = U.load(f"snv_train_{tag}.npz")
model, _ = product_from_hp(fx["hp"])
orc = U.snv_oracle_from_hp(fx["hp"], drops=(0.0, 0.0, 0.0))
model.load_state_dict(U.snv_state_for(fx, orc))
for m in model.modules():
    if isinstance(m, nn.Dropout):
        m.p = 0.0
model = model.cuda().train()
orig = T._call


def c64(t):
    return None if t is None else t.detach().cpu().double()


def checked(name, *args):
    if name == "mural_op_conv32_wgrad":
        dy, x, B, L, s, t, relu, dW, db = args[:9]
        pre = {k: c64(v) for k, v in dict(dy=dy, x=x, s=s, t=t).items()}
        orig(name, *args)
        torch.cuda.synchronize()
        xin = F.relu(pre["x"]) if relu else pre["x"]
        xin = pre["s"].view(1, -1, 1) * xin + pre["t"].view(1, -1, 1)
        with torch.enable_grad():
            W = torch.zeros(32, 32, 3, dtype=torch.float64, requires_grad=True)
            F.conv1d(xin, W, None, padding=1).backward(pre["dy"])
        e = ((c64(dW) - W.grad).abs().max() / (W.grad.abs().max() + 1e-9)).item()
        same = (c64(x) - pre["x"]).abs().max().item()
        print("wgrad B=%d L=%d relerr %.2e  (x changed by the call: %.1e)  nan=%s" % (B, L, e, same, bool(torch.isnan(c64(dW)).any())))
        return
    if name == "mural_op_conv32" and args[13] == 2:
        dy, W, _, dz, B, L = args[:6]
        relu, sx, mean, invstd, acc = args[14:19]
        orig(name, *args)
        torch.cuda.synchronize()
        r = F.relu(c64(sx)) if relu else c64(sx)
        m_ref = r.mean((0, 2))
        i_ref = 1.0 / torch.sqrt(r.var((0, 2), unbiased=False) + 1e-5)
        print("dgrad+sums B=%d L=%d: mean err %.2e invstd err %.2e" % (B, L, (c64(mean) - m_ref).abs().max().item(),
                                                                      ((c64(invstd) - i_ref).abs() / i_ref).max().item()))
        return
    orig(name, *args)


T._call = checked
cat = torch.from_numpy(fx["cat"]).cuda()
x = U.onehot(fx["codes"]).cuda()
preds = model((torch.zeros(len(cat), 1, dtype=torch.float64, device="cuda"), cat), x)
loss = nn.CrossEntropyLoss(reduction="sum")(preds, torch.from_numpy(fx["y"]).cuda())
model.zero_grad()
loss.backward()
worst = ("", 0.0)
for k, p in model.named_parameters():
    if ".layer." in k or p.numel() == 0:
        continue
    want = fx["g::" + k]
    err = float(np.abs(p.grad.cpu().numpy() - want).max()) / (float(np.abs(want).max()) + 1e-2)
    if err > worst[1]:
        worst = (k, err)
print("worst gradient error with per-call syncs:", worst)
