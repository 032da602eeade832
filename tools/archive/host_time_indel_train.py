"""GPU box: where the host time of the INDEL training step goes (forward / loss / backward / clip / optimizer), no device syncs
inside the loop.  usage: python tools/archive/host_time_indel_train.py [batch]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mural_amd.model import model_choice, weights_init  # noqa: E402
from mural_amd.train import clip_grad_norm_  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 16
cfg = dict(CNN_out_channels=8, CNN_kernel_size=7, down_list=[1, 4, 5, 5, 5, 2], use_reverse=True)
torch.manual_seed(0)
model = model_choice(0, cfg, dict(n_class=8), "indel")
model.apply(weights_init)
model = model.cuda().train()
opt = torch.optim.Adam(model.parameters(), lr=1e-3, fused=True)
crit = torch.nn.CrossEntropyLoss(reduction="sum")
codes = torch.randint(0, 4, (B, 8000), device="cuda")
x = torch.nn.functional.one_hot(codes, 4).permute(0, 2, 1).float().contiguous()
y = torch.randint(0, 8, (B,), device="cuda")
acc = {}


def lap(k, t0):
    t = time.perf_counter()
    acc[k] = acc.get(k, 0.0) + t - t0
    return t


def step():
    t = time.perf_counter()
    out = model(x)
    t = lap("forward", t)
    loss = crit(out, y)
    t = lap("loss", t)
    opt.zero_grad()
    t = lap("zero_grad", t)
    loss.backward()
    t = lap("backward", t)
    clip_grad_norm_(model, 10)
    t = lap("clip", t)
    opt.step()
    lap("optimizer", t)


for _ in range(5):
    step()
if "--no-freeze" not in sys.argv:
    from mural_amd.train import freeze_host_heap
    freeze_host_heap()    # as mural_amd.train.train_epoch does
torch.cuda.synchronize()
acc.clear()
n = 30
t0 = time.perf_counter()
for _ in range(n):
    step()
th = time.perf_counter() - t0
torch.cuda.synchronize()
print("batch %d: host %.2f ms/step, with the device drained %.2f ms/step" % (B, th / n * 1e3, (time.perf_counter() - t0) / n * 1e3))
for k, v in acc.items():
    print("  %-10s %.2f ms" % (k, v / n * 1e3))
