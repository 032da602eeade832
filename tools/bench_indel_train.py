"""GPU-box measurement of the INDEL (UNet_Small) training step: forward (batch-statistics BatchNorm, dropout) + CE(sum) +
backward + clip_grad_norm_ + Adam, human insertion geometry (L = 8000, 8 classes, use_reverse).
usage: python tools/bench_indel_train.py [batch] [steps]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mural_amd.model import model_choice, weights_init  # noqa: E402
from mural_amd.train import clip_grad_norm_  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
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


def step():
    loss = crit(model(x), y)
    opt.zero_grad()
    loss.backward()
    clip_grad_norm_(model, 10)
    opt.step()
    return loss


for _ in range(3):
    step()
from mural_amd.train import freeze_host_heap  # noqa: E402
freeze_host_heap()        # as mural_amd.train.train_epoch does
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    loss = step()
t_host = (time.perf_counter() - t0) / steps
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print("host enqueue %.2f ms/step (the device drains the queue in %.2f ms/step)" % (t_host * 1e3, dt * 1e3))
# forward 113.4 MFLOP/position; backward = input + weight gradients of every conv ~ 2x (the first conv has no input gradient)
print("UNet_Small train step B=%d L=8000: %.1f ms/step = %.1f positions/s; loss %.3f; %.1f TFLOP/s algorithmic (3 x 113.4 MFLOP/pos)"
      % (B, dt * 1e3, B / dt, loss.item(), B / dt * 3 * 113.4e6 / 1e12))

if "--cpu" in sys.argv:      # the CPU restatement (oracle, = the reference's torch modules) on the host cores, one bounded step
    from oracle import indel_ref
    cb = 16
    orc = indel_ref.build(n_class=8, channels=8, ksize=7, down_list=(1, 4, 5, 5, 5, 2), use_reverse=True).train()
    oc = torch.optim.Adam(orc.parameters(), lr=1e-3)
    xc, yc = x[:cb].cpu(), y[:cb].cpu()
    for i in range(3):
        t0 = time.perf_counter()
        l = crit(orc(xc), yc)
        oc.zero_grad()
        l.backward()
        torch.nn.utils.clip_grad_norm_(orc.parameters(), 10)
        oc.step()
        dt = time.perf_counter() - t0
    print("CPU oracle train step B=%d: %.1f ms = %.1f positions/s on %d threads" % (cb, dt * 1e3, cb / dt, torch.get_num_threads()))
