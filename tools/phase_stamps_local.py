"""GPU box: phases of the fused local branch's backward launches inside one training step at batch 4096 (mural_debug_lt_set_stamps)."""
import os
os.environ.setdefault("MURAL_HIP_FLAVOR", "debug")      # validation hooks / development switches: the debug flavour of the library
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from mural_amd import _lib  # noqa: E402
from mural_amd.data import PackedGenome  # noqa: E402
from mural_amd.train import CrossEntropySum  # noqa: E402

dev = torch.device("cuda", 0)
B = 4096
codes = bench.synthetic_genome(1_000_000 + 2000)
packed, mask = bench.pack2(codes)
genome = PackedGenome(packed, mask, len(codes), dev)
model = bench.build_model(dev).train()
crit = CrossEntropySum()
idx = torch.arange(B, device=dev)
pos, strand = idx + 1000, (idx & 1).to(torch.uint8)
cat = genome.encode_kmer(pos, strand, 10, 3)
x = genome.encode_symbols(pos, strand, 1000)
y = torch.zeros(B, dtype=torch.int64, device=dev)
cont = torch.zeros(B, 1, device=dev)
stamps = torch.zeros(3 * 256 * 8, dtype=torch.int64, device=dev)
for it in range(4):
    if it == 3:
        _lib.check(_lib.lib().mural_debug_lt_set_stamps(stamps.data_ptr()))
    loss = crit(model((cont, cat), x), y)
    model.zero_grad()
    loss.backward()
    torch.cuda.synchronize()
_lib.check(_lib.lib().mural_debug_lt_set_stamps(None))
s = stamps.view(3, 256, 8).cpu().double()
names = ["entry", "prologue", "A ready", "first block's MFMAs", "blocks done", "barrier", "exit"]
for k, nm in enumerate(("B3 (top)", "B2", "B1 (bottom)")):
    t = s[k]
    t0 = t[:, 0][t[:, 0] > 0].min()
    us = (t - t0) / 100.0
    cols = [c for c in range(7) if (t[:, c] > 0).any()]
    print(nm + ": " + "; ".join("%s %.1f..%.1f (mean %.1f)" % (names[c], us[:, c].min(), us[:, c].max(), us[:, c].mean()) for c in cols))
