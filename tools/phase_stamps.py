"""Diagnostic (GPU box): per-phase cycle shares of wave 0 in the fused tower kernel (s_memtime stamps)."""
import os
os.environ.setdefault("MURAL_HIP_FLAVOR", "debug")      # validation hooks / development switches: the debug flavour of the library
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from mural_amd import _lib  # noqa: E402
from mural_amd.data import PackedGenome  # noqa: E402

dev = torch.device("cuda", 0)
codes = bench.synthetic_genome(2_000_000 + 2000)
packed, mask = bench.pack2(codes)
genome = PackedGenome(packed, mask, len(codes), dev)
model = bench.build_model(dev)
B = 200_000
idx = torch.arange(B, device=dev, dtype=torch.int64)
pos, strand = idx + 1000, (idx & 1).to(torch.uint8)
with torch.no_grad():
    model.forward_packed(genome, pos, strand, 10, 3)
    torch.cuda.synchronize()
    stamps = torch.zeros(2048 * 32, dtype=torch.int64, device=dev)
    _lib.check(_lib.lib().mural_debug_set_stamps(stamps.data_ptr()))
    model.forward_packed(genome, pos, strand, 10, 3)
    torch.cuda.synchronize()
    _lib.check(_lib.lib().mural_debug_set_stamps(None))
s = stamps.view(2048, 32).double().cpu()[:int(os.environ.get("MURAL_DEBUG_TOWER_GRID", "2048"))]
grid = int(os.environ.get("MURAL_DEBUG_TOWER_GRID", "2048"))
tiles_per_block = (32768 / 2) / grid * (B / 32768)
mean = s.mean(dim=0) / tiles_per_block
names = {0: "decode", 25: "fc", 26: "head"}
for tw, nm in ((0, "L"), (1, "M")):
    for k, v in {1: "kidx+LUTload", 2: "stage1", 3: "xres", 4: "barrier s2", 5: "pool2", 6: "barrier s3", 7: "pool3",
                 8: "barrier s4", 9: "conv", 10: "gmax"}.items():
        names[k + 12 * tw] = nm + ":" + v
tot = float(mean.sum())
print("cycles (s_memtime ticks, 100 MHz) per tile for wave 0; total %.1f" % tot)
for i in range(32):
    if mean[i] > 0:
        print("  %-16s %9.1f  %5.1f%%" % (names.get(i, str(i)), float(mean[i]), 100 * float(mean[i]) / tot))
