"""GPU box: the INDEL training step captured into a HIP graph (torch.cuda.CUDAGraph) next to the eager step."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mural_amd.model import model_choice, weights_init  # noqa: E402
from mural_amd.model import train_ops as T  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
cfg = dict(CNN_out_channels=8, CNN_kernel_size=7, down_list=[1, 4, 5, 5, 5, 2], use_reverse=True)
torch.manual_seed(0)
model = model_choice(0, cfg, dict(n_class=8), "indel")
model.apply(weights_init)
model = model.cuda().train()
opt = torch.optim.Adam(model.parameters(), lr=1e-3, capturable=True, fused=True)
crit = torch.nn.CrossEntropyLoss(reduction="sum")
codes = torch.randint(0, 4, (B, 8000), device="cuda")
x = torch.nn.functional.one_hot(codes, 4).permute(0, 2, 1).float().contiguous()
y = torch.randint(0, 8, (B,), device="cuda")
seed = torch.zeros(1, dtype=torch.int64, device="cuda")
T.set_device_seed(seed)
loss_box = [None]


def step():
    seed.add_(0x9E3779B97F4A7C15 & 0x7FFFFFFFFFFFFFFF)
    loss = crit(model(x), y)
    opt.zero_grad(set_to_none=False)
    loss.backward()
    torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=10, error_if_nonfinite=False)
    opt.step()
    loss_box[0] = loss


side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3):
        step()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    step()
torch.cuda.synchronize()
print("eager  : %.2f ms/step" % ((time.perf_counter() - t0) / 20 * 1e3))
T.reset_zero_arena()
T.captured_status.clear()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    step()
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    g.replay()
torch.cuda.synchronize()
print("graphed: %.2f ms/step, loss %.3f" % ((time.perf_counter() - t0) / 20 * 1e3, loss_box[0].item()))
