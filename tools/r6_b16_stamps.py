"""GPU box: phase stamps (s_memtime, 100 MHz) of the tower-parallel small-call kernel for a 16-site packed call: where its 36 us go."""
import os
os.environ.setdefault("MURAL_HIP_FLAVOR", "debug")
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from mural_amd import _lib  # noqa: E402
from mural_amd.data import PackedGenome  # noqa: E402

dev = torch.device("cuda", 0)
codes = bench.synthetic_genome(200_000 + 2000)
packed, mask = bench.pack2(codes)
genome = PackedGenome(packed, mask, len(codes), dev)
model = bench.build_model(dev)
idx = torch.arange(16, device=dev, dtype=torch.int64) * 101
pos, strand = idx + 1000, (idx & 1).to(torch.uint8)
with torch.no_grad():
    for _ in range(5):
        model.forward_packed(genome, pos, strand, 10, 3)
    torch.cuda.synchronize()
    stamps = torch.zeros(2048 * 32, dtype=torch.int64, device=dev)
    _lib.check(_lib.lib().mural_debug_set_stamps(stamps.data_ptr()))
    model.forward_packed(genome, pos, strand, 10, 3)
    torch.cuda.synchronize()
    _lib.check(_lib.lib().mural_debug_set_stamps(None))
s = stamps.view(2048, 32).cpu()
for wg in (0, 1, 2, 3):
    row = s[wg].tolist()
    print("wg", wg, "ticks (10 ns):", [int(v) for v in row[:28]], "sum", sum(row[:28]))
