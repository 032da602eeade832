"""Diagnostic: per-parameter gradient differences of the INDEL training step vs the golden reference step."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import _util as U
from tests.test_gpu_indel import product_from

for tag in ("rev", "norev"):
    fx = U.load(f"indel_train_{tag}.npz")
    model = product_from(fx)
    orc = U.indel_oracle_from_hp(fx["hp"], fx["down"])
    model.load_state_dict(U.indel_state_for(fx, orc), strict=True)
    model = model.cuda().train()
    model.out_fc[1].p = 0.0
    preds = model(U.onehot(fx["codes"]).cuda())
    loss = torch.nn.CrossEntropyLoss(reduction="sum")(preds, torch.from_numpy(fx["y"]).cuda())
    loss.backward()
    for k, p in model.named_parameters():
        want = fx["g::" + k]
        d = np.abs(p.grad.cpu().numpy() - want).max()
        print(f"{tag} {k:34s} max|want| {np.abs(want).max():.3e}  diff {d:.3e}  rel {d / (np.abs(want).max() + 1e-12):.2e}")

# self-consistency of every conv backward call: db vs dy.sum, dW / dx vs torch's conv gradients on the same dy
from mural_amd.model import train_ops as T
orig_call = T._call


def checked(name, *args):
    orig_call(name, *args)
    if name != "mural_op_convg_bwd":
        return
    dy, x, weight, B, Cin, Lin, Cout, K, stride, pad, up, dx, dW, db = args[:14]
    with torch.enable_grad():
        xr = x.detach().clone().requires_grad_()
        wr = weight.detach().clone().requires_grad_()
        xu = xr.repeat_interleave(up, dim=2) if up > 1 else xr
        y = torch.nn.functional.conv1d(xu, wr, None, stride=stride, padding=pad)
        y.backward(dy)
    msg = f"conv {tuple(weight.shape)} s{stride} up{up} L{Lin}: dW rel {float((dW - wr.grad).abs().max() / (wr.grad.abs().max() + 1e-12)):.2e}"
    if dx is not None:
        msg += f" dx rel {float((dx - xr.grad).abs().max() / (xr.grad.abs().max() + 1e-12)):.2e}"
    if db is not None:
        want = dy.sum((0, 2))
        msg += f" db abs {float((db - want).abs().max()):.2e} of {float(want.abs().max()):.2e}"
    print(msg)


T._call = checked
fx = U.load("indel_train_norev.npz")
model = product_from(fx)
orc = U.indel_oracle_from_hp(fx["hp"], fx["down"])
model.load_state_dict(U.indel_state_for(fx, orc), strict=True)
model = model.cuda().train()
model.out_fc[1].p = 0.0
preds = model(U.onehot(fx["codes"]).cuda())
loss = torch.nn.CrossEntropyLoss(reduction="sum")(preds, torch.from_numpy(fx["y"]).cuda())
loss.backward()
