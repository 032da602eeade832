"""Long windows, more sites than one chunk (8192): the second chunk's outputs against the same sites run alone."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
device = torch.device("cuda", 0)
R4 = 2000
g = torch.Generator(device=device).manual_seed(5)
m4 = bench.build_model(device, R4)
B = 8192 + 700
codes = torch.randint(0, 4, (B, 2 * R4 + 1), device=device, generator=g)
x = torch.nn.functional.one_hot(codes, 4).permute(0, 2, 1).float().contiguous()
c = codes[:, R4 - bench.LOCAL_RADIUS:R4 + bench.LOCAL_RADIUS + 1]
cat = (c[:, :-2] * 16 + c[:, 1:-1] * 4 + c[:, 2:]).contiguous()
cont = torch.zeros(B, 1, device=device, dtype=torch.float64)
with torch.no_grad():
    full = m4((cont, cat), x)
    tail = m4((cont[8192:], cat[8192:]), x[8192:].contiguous())
    head = m4((cont[:500], cat[:500]), x[:500].contiguous())
print("fused:", m4._fused_ok(), " second chunk max diff %.3e, first chunk max diff %.3e" %
      (float((full[8192:] - tail).abs().max()), float((full[:500] - head).abs().max())))
