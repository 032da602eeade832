"""Diagnostic (GPU box): per-phase cycle shares of wave 0 of every workgroup in the wave-private tower kernel (s_memtime stamps)."""
import os
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
B = 131072 * 2
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
s = stamps.view(2048, 32).double().cpu()[:512]
mean = s.mean(dim=0)
names = {1: "entry", 2: "convs", 3: "last conv", 4: "pool2+store", 9: "s3 entry", 10: "convs", 11: "pool3", 12: "last conv", 13: "gmax/fc/head"}
for tw, nm in ((0, "large"), (1, "mid")):
    for base, ph in ((0, "first stage"), (8, "short stages")):
        seg = mean[16 * tw + base:16 * tw + base + 8]
        tot = float(seg.sum())
        if tot <= 0:
            continue
        print(f"{nm} {ph}: {tot:.0f} ticks per wave over the two chunks")
        for i in range(8):
            if seg[i] > 0:
                print("   %-14s %12.0f  %5.1f%%" % (names.get(base + i, str(i)), float(seg[i]), 100 * float(seg[i]) / tot))

raw = stamps.view(2048, 32).cpu()[:512]
keys, cnts, hw = raw[:, 30].tolist(), raw[:, 31].tolist(), raw[:, 29].tolist()
from collections import Counter
print("distinct CU keys:", len(set(keys)), " workgroups per key:", Counter(Counter(keys).values()))
print("first 12 (block, key, arrival, hw_id):", [(i, hex(keys[i]), cnts[i], hex(hw[i])) for i in range(12)])
print("blocks 256..262:", [(i, hex(keys[i]), cnts[i], hex(hw[i])) for i in range(256, 262)])
par = Counter((k, c & 1) for k, c in zip(keys, cnts))
print("keys with both parities:", sum(1 for k in set(keys) if par[(k, 0)] and par[(k, 1)]))
