"""GPU-box measurement of the INDEL (UNet_Small) forward: positions/s for the human insertion geometry (L=8000)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mural_amd.model import model_choice, weights_init  # noqa: E402

cfg = dict(CNN_out_channels=8, CNN_kernel_size=7, down_list=[1, 4, 5, 5, 5, 2], use_reverse=True)
torch.manual_seed(0)
model = model_choice(0, cfg, dict(n_class=8), "indel")
model.apply(weights_init)
model = model.cuda().eval()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
MODE = sys.argv[2] if len(sys.argv) > 2 else "both"      # dense | packed | both
if MODE != "packed":      # (the packed entry needs no dense windows: their generation would sit in its kernel traces and counters)
    codes = torch.randint(0, 4, (B, 8000), device="cuda")
    x = torch.nn.functional.one_hot(codes, 4).permute(0, 2, 1).float().contiguous()
with torch.no_grad():
    for _ in range(2 if MODE != "packed" else 0):
        model(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5 if MODE != "packed" else 0):
        model(x)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
if MODE != "packed":
    print("UNet_Small insertion L=8000: %.1f positions/s (%.2f ms per %d), %.1f TFLOP/s algorithmic (113.4 MFLOP/pos)"
          % (B / dt, dt * 1e3, B, B / dt * 113.4e6 / 1e12))
if MODE == "dense":
    sys.exit(0)

# the packed entry (window decode + input conv inside the first level's kernel)
from mural_amd.data import PackedGenome  # noqa: E402
rng = np.random.default_rng(0)
seq = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, size=2_000_000)].tobytes().decode()
genome = PackedGenome.from_sequence(seq, "cuda")
idx = torch.arange(B, device="cuda", dtype=torch.int64)
pos, strand = idx * 47 + 4000, (idx & 1).to(torch.uint8)
with torch.no_grad():
    for _ in range(2):
        model.forward_packed(genome, pos, strand, 4000)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        model.forward_packed(genome, pos, strand, 4000)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
print("packed entry: %.1f positions/s (%.2f ms per %d), %.1f TFLOP/s algorithmic" % (B / dt, dt * 1e3, B, B / dt * 113.4e6 / 1e12))
