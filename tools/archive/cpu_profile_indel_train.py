"""GPU box: host-side profile (cProfile) of the INDEL training step (tools/bench_indel_train.py's loop at batch 16)."""
import cProfile
import os
import pstats
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mural_amd.model import model_choice, weights_init  # noqa: E402

B = 16
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
    torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=10, error_if_nonfinite=False)
    opt.step()


for _ in range(5):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(30):
    step()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(35)
