"""GPU-box debugging aid: per-parameter gradient errors of the training step vs the reference's golden vectors."""
import os
import sys

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import _util as U  # noqa: E402
from tests.test_gpu_snv import product_from_hp  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "T"
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
print("preds err", np.abs(preds.detach().cpu().numpy() - fx["preds"]).max())
loss = nn.CrossEntropyLoss(reduction="sum")(preds, torch.from_numpy(fx["y"]).cuda())
model.zero_grad()
loss.backward()
print("loss", loss.item(), float(fx["loss"]))
rows = []
for k, p in model.named_parameters():
    if ".layer." in k or p.numel() == 0:
        continue
    want = fx["g::" + k]
    scale = max(float(np.abs(want).max()), 1e-12)
    err = float(np.abs(p.grad.cpu().numpy() - want).max()) / scale
    rows.append((err, k, scale))
for err, k, scale in rows:
    print(f"{err:10.3e}  {k:40s} max|g|={scale:.3e}")
for k, b in model.named_buffers():
    if ".layer." in k or k.endswith("num_batches_tracked") or b.numel() == 0:
        continue
    e = np.abs(b.cpu().numpy() - fx["b::" + k]).max()
    if e > 2e-5:
        print("buffer", k, e)
