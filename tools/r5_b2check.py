import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch, torch.nn as nn
from tests import _util as U
from tests.test_gpu_snv import product_from_hp
OPS = {"MURAL_TRAIN_LOCAL_OPS": "1", "MURAL_TRAIN_HEAD_OPS": "1", "MURAL_TRAIN_FIRST_SEPARATE": "1", "MURAL_TRAIN_NO_FIRST_FOLD": "1", "MURAL_DEBUG_FIRST_SCATTER": "1"}
hp = np.array((5, 3, 100, 150, 75, 32, 3, 2, 2)); B = 2
rng = np.random.default_rng(int(hp.sum()) + B)
r, order, R = 5, 3, 100
ncol = 2 * r + 1 - (order - 1)
cat = torch.from_numpy(rng.integers(0, 65, size=(B, ncol)))
codes = rng.integers(0, 4, size=(B, 2 * R + 1)).astype(np.uint8)
x = U.onehot(codes)
y = torch.from_numpy(rng.integers(0, 2, size=B))
orc = U.snv_oracle_from_hp(hp, drops=(0.0, 0.0, 0.0))
sd = {k: v.clone() for k, v in orc.state_dict().items()}
orc64 = U.snv_oracle_from_hp(hp, drops=(0.0, 0.0, 0.0)).double()
orc64.load_state_dict({k: v.double() if v.is_floating_point() else v for k, v in sd.items()})
orc64.train()
out = orc64((torch.zeros(B, 1, dtype=torch.float64), cat), x.double())
nn.CrossEntropyLoss(reduction="sum")(out, y).backward()
ref = {k: p.grad.clone() for k, p in orc64.named_parameters() if p.numel() and p.grad is not None}
for ops in (False, True):
    for k, v in OPS.items():
        if ops: os.environ[k] = v
        else: os.environ.pop(k, None)
    model, _ = product_from_hp(hp)
    model.load_state_dict(sd)
    for m in model.modules():
        if isinstance(m, nn.Dropout): m.p = 0.0
    model = model.cuda().train()
    o = model((torch.zeros(B, 1, device="cuda"), cat.cuda()), x.cuda())
    nn.CrossEntropyLoss(reduction="sum")(o, y.cuda()).backward()
    worst = max(((float((p.grad.cpu().double() - ref[k]).abs().max()) / (float(ref[k].abs().max()) + 1e-2), k) for k, p in model.named_parameters() if k in ref and ".layer." not in k))
    print("ops" if ops else "fused", "worst rel grad err vs float64 oracle:", worst, "out err", float((o.cpu().double() - out).abs().max()))
