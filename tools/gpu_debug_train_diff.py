"""GPU debugging aid: run one training step twice with different conv32 tile heights (MURAL_DEBUG_CONV32_R) and report the
first kernel call whose tensor arguments differ."""
import os
import sys

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import _util as U  # noqa: E402
from tests.test_gpu_snv import product_from_hp  # noqa: E402
from mural_amd.model import train_ops as T  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "S"
fx = U.load(f"snv_train_{tag}.npz")
orig = T._call
log = []


def logged(name, *args):
    orig(name, *args)
    torch.cuda.synchronize()
    log.append((name, [a.detach().cpu().clone() if isinstance(a, torch.Tensor) else a for a in args]))


T._call = logged


def run(r):
    os.environ["MURAL_DEBUG_CONV32_R"] = str(r)
    log.clear()
    model, _ = product_from_hp(fx["hp"])
    orc = U.snv_oracle_from_hp(fx["hp"], drops=(0.0, 0.0, 0.0))
    model.load_state_dict(U.snv_state_for(fx, orc))
    for m in model.modules():
        if isinstance(m, nn.Dropout):
            m.p = 0.0
    model = model.cuda().train()
    cat = torch.from_numpy(fx["cat"]).cuda()
    x = U.onehot(fx["codes"]).cuda()
    preds = model((torch.zeros(len(cat), 1, dtype=torch.float64, device="cuda"), cat), x)
    loss = nn.CrossEntropyLoss(reduction="sum")(preds, torch.from_numpy(fx["y"]).cuda())
    model.zero_grad()
    loss.backward()
    return list(log)


a, b = run(0), run(1)     # 0: default tile height, 1: one row per tile
print(len(a), len(b), "calls")
for i, ((na, aa), (nb_, ab)) in enumerate(zip(a, b)):
    assert na == nb_
    for j, (u, v) in enumerate(zip(aa, ab)):
        if na == 'mural_op_conv32_wgrad' and j == 9:
            continue
        if isinstance(u, torch.Tensor) and u.dtype == torch.int32 and u.shape == v.shape and (u != v).any():
            print("  note: call %d %s int arg %d differs in %d of %d entries" % (i, na, j, int((u != v).sum()), u.numel()))
        if isinstance(u, torch.Tensor) and u.numel() and u.dtype in (torch.float32, torch.float64) and u.shape == v.shape:
            if u.dim() == 3 and u.shape[0] == 32 and u.dtype == torch.float64:
                u, v = u.sum(0), v.sum(0)
            d = (u.double() - v.double()).abs().max().item()
            s = u.double().abs().max().item() + 1e-12
            if d / s > 1e-5 and na in ("mural_op_maxpool_fwd", "mural_op_maxpool_bwd"):
                print("  note: call %d %s arg %d differs by rel %.2e" % (i, na, j, d / s))
            if d / s > 1e-3:
                ints = [x for x in aa if isinstance(x, int)]
                print("call %d %s arg %d differs: rel %.2e (max %.2e) ints=%s" % (i, na, j, d / s, s, ints[:8]))
                break
    else:
        continue
    break
