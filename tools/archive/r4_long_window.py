"""The long-window variant of bench.py alone (distal_radius 4000, 512 windows per call): python tools/archive/r4_long_window.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
device = torch.device("cuda", 0)
R4 = 4000
g = torch.Generator(device=device).manual_seed(3)
m4 = bench.build_model(device, R4)
B4 = int(sys.argv[1]) if len(sys.argv) > 1 else 512
codes4 = torch.randint(0, 4, (B4, 2 * R4 + 1), device=device, generator=g)
x4 = torch.nn.functional.one_hot(codes4, 4).permute(0, 2, 1).float().contiguous()
c4 = codes4[:, R4 - bench.LOCAL_RADIUS:R4 + bench.LOCAL_RADIUS + 1]
cat4 = (c4[:, :-2] * 16 + c4[:, 1:-1] * 4 + c4[:, 2:]).contiguous()
cont4 = torch.zeros(B4, 1, device=device, dtype=torch.float64)
with torch.no_grad():
    for _ in range(3):
        m4((cont4, cat4), x4)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        m4((cont4, cat4), x4)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
fl = bench.snv_flop_per_site(R4)
print("long window R=4000: %.2f ms per %d windows = %.0f bases/s, %.2f TFLOP/s (%.3f of the fp32 MFMA roof)"
      % (dt * 1e3, B4, B4 / dt, fl * B4 / dt / 1e12, fl * B4 / dt / 1e12 / bench.PEAK_FP32_MFMA_TFLOPS))

# fused (segmented) against the per-layer path on the same inputs, more sites than one launch's resident waves
import numpy as np
B5 = 3000
codes5 = torch.randint(0, 4, (B5, 2 * R4 + 1), device=device, generator=g)
x5 = torch.nn.functional.one_hot(codes5, 4).permute(0, 2, 1).float().contiguous()
c5 = codes5[:, R4 - bench.LOCAL_RADIUS:R4 + bench.LOCAL_RADIUS + 1]
cat5 = (c5[:, :-2] * 16 + c5[:, 1:-1] * 4 + c5[:, 2:]).contiguous()
cont5 = torch.zeros(B5, 1, device=device, dtype=torch.float64)
with torch.no_grad():
    a = m4((cont5, cat5), x5).cpu().numpy()
from mural_amd.model import generic_eval
from mural_amd.model.model_snv import POOLS_MID, POOLS_LARGE
with torch.no_grad():
    b = generic_eval.forward(m4, cat5, x5, POOLS_MID, POOLS_LARGE).cpu().numpy()
print("fused ok:", m4._fused_ok(), " max |fused - per-layer| over %d sites: %.3e (log-probabilities)" % (B5, np.abs(a - b).max()))
