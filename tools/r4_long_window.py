"""The long-window variant of bench.py alone (distal_radius 4000, 512 windows per call): python tools/r4_long_window.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
device = torch.device("cuda", 0)
R4 = 4000
g = torch.Generator(device=device).manual_seed(3)
m4 = bench.build_model(device, R4)
B4 = 512
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
