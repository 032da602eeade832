"""GPU box: per-tensor distance of one training step from a G7 fixture (tests/golden/snv_train_<tag>.npz), both conv kernel families."""
import os
import sys

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import _util as U  # noqa: E402
from tests.test_gpu_snv import product_from_hp  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "S"
fx = U.load(f"snv_train_{tag}.npz")
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
loss.backward()
print("preds", np.abs(preds.detach().cpu().numpy() - fx["preds"]).max(), "loss", loss.item(), float(fx["loss"]))
for k, p in model.named_parameters():
    if ".layer." in k or p.numel() == 0:
        continue
    want = fx["g::" + k]
    g = p.grad.cpu().numpy()
    scale = float(np.abs(want).max()) + 1e-2
    err = float(np.abs(g - want).max()) / scale
    if err > 5e-5:
        print(f"{k:28s} |want| {np.abs(want).max():.3e} err {err:.2e}  abs {np.abs(g - want).max():.3e}")
