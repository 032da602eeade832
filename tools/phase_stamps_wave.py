"""Diagnostic (GPU box): per-phase cycle shares of wave 0 of every workgroup in the wave-private tower kernel (s_memtime stamps)."""
import os
os.environ.setdefault("MURAL_HIP_FLAVOR", "debug")      # validation hooks / development switches: the debug flavour of the library
import sys

import torch
from collections import Counter

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
names = {1: "entry", 2: "convs", 3: "last conv", 4: "pool2+store", 9: "s3 entry", 10: "convs", 11: "pool3", 12: "last conv", 13: "gmax/fc/head", 14: "s3 tile -> image", 15: "zero gaps"}
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

# wall-clock life of every wave of the last launch of each (phase, tower): how much of the launch the average wave is alive
wl = stamps.view(2048, 32)[1024:].cpu().double()
for q, nm in enumerate(("large first", "mid first", "large short", "mid short")):
    st = torch.cat([wl[:, 4 * q], wl[:, 4 * q + 2]])
    en = torch.cat([wl[:, 4 * q + 1], wl[:, 4 * q + 3]])
    ok = st > 0
    st, en = st[ok], en[ok]
    if len(st) == 0:
        continue
    span = float(en.max() - st.min())
    life = en - st
    qs = torch.quantile((en - st.min()) / span, torch.tensor([0.0, 0.1, 0.5, 0.9, 1.0], dtype=torch.double))
    print("%-12s waves %d  launch span %.1f us  mean life %.3f of span  start spread %.1f us  end quantiles (0/10/50/90/100%%): %s"
          % (nm, len(st), span / 100, float(life.mean()) / span, float(st.max() - st.min()) / 100, [round(float(v), 3) for v in qs]))

# who is slow?  end time of the large first-stage launch by XCC / SE / CU / SIMD / wave slot (HW_ID: wave [3:0], simd [5:4], cu [11:8], sh [12], se [15:13])
hw = torch.cat([wl[:, 16], wl[:, 18]]).long()
xc = torch.cat([wl[:, 17], wl[:, 19]]).long()
st = torch.cat([wl[:, 0], wl[:, 2]])
en = torch.cat([wl[:, 1], wl[:, 3]])
rel = (en - st.min()) / float(en.max() - st.min())
for nm, key in (("xcc", xc & 7), ("se", (hw >> 13) & 7), ("cu", (hw >> 8) & 15), ("simd", (hw >> 4) & 3), ("wave slot", hw & 15)):
    vals = sorted(set(key.tolist()))
    print(nm, " ".join("%d:%.3f(n=%d)" % (v, float(rel[key == v].mean()), int((key == v).sum())) for v in vals))
cuid = (xc & 7) * 4096 + ((hw >> 13) & 7) * 256 + ((hw >> 12) & 1) * 16 + ((hw >> 8) & 15)
per_cu = {}
for c, s_, r in zip(cuid.tolist(), ((hw >> 4) & 3).tolist(), rel.tolist()):
    per_cu.setdefault(c, []).append((s_, round(r, 3)))
print("distinct CUs:", len(per_cu), "waves per CU:", Counter(len(v) for v in per_cu.values()))
print("waves per (CU, SIMD):", Counter(Counter((c, s_) for c, s_ in zip(cuid.tolist(), ((hw >> 4) & 3).tolist())).values()))
for c in list(per_cu)[:6]:
    print("  cu %05x:" % c, sorted(per_cu[c]))

raw = stamps.view(2048, 32).cpu()[:512]
keys, cnts, hw = raw[:, 30].tolist(), raw[:, 31].tolist(), raw[:, 29].tolist()
print("distinct CU keys:", len(set(keys)), " workgroups per key:", Counter(Counter(keys).values()))
print("first 12 (block, key, arrival, hw_id):", [(i, hex(keys[i]), cnts[i], hex(hw[i])) for i in range(12)])
print("blocks 256..262:", [(i, hex(keys[i]), cnts[i], hex(hw[i])) for i in range(256, 262)])
par = Counter((k, c & 1) for k, c in zip(keys, cnts))
print("keys with both parities:", sum(1 for k in set(keys) if par[(k, 0)] and par[(k, 1)]))
