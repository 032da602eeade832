"""Diagnostic: phase times of the split-form 8-channel ConvBlock (conv1d.hip, convblock_kernel<8, ., ., true>) at the bench geometry
(2048 rows of 8000): per workgroup the first thread's clock at entry / front input staged / block input ready / SiLU done / block
output ready / exit (mural_debug_cb8_set_stamps), averaged; and the number of workgroups alive at a time."""
import os
os.environ.setdefault("MURAL_HIP_FLAVOR", "debug")      # validation hooks / development switches: the debug flavour of the library
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mural_amd import _lib
import tools.gpu_debug_convblock as G      # noqa: E402

lib = _lib.lib()
for front, skip, tail in (((4, 1), False, False), ((16, 4), True, True), (None, False, False)):
    _, call = G.run(1, 2048, 8, 8000, front, skip, tail, seed=1)
    for _ in range(2):
        call()
    torch.cuda.synchronize()
    stamps = torch.zeros(8 * 65536, dtype=torch.int64, device="cuda")
    lib.mural_debug_cb8_set_stamps(stamps.data_ptr())
    call()
    torch.cuda.synchronize()
    lib.mural_debug_cb8_set_stamps(None)
    s = stamps.view(-1, 8).cpu().double()
    s = s[s[:, 0] > 0]
    t0 = s[:, 0].min()
    names = ["front input staged", "block input ready", "k=5 + SiLU", "1x1 + loads", "tail / store"]
    idx = [(0, 1), (1, 2), (2, 3), (3, 4), (4, 5)] if front else [(0, 2), (0, 2), (2, 3), (3, 4), (4, 5)]
    print("front", front, "skip", skip, "tail", tail, ": %d workgroups stamped, launch span %.1f us" % (len(s), (s[:, 5].max() - t0) * 0.01))
    for n, (a, b) in zip(names, idx):
        if not front and n == "front input staged":
            continue
        print("   %-20s %.2f us" % (n, (s[:, b] - s[:, a]).mean() * 0.01))
    if not tail:
        print("   %-20s %.2f us" % ("stores drained", (s[:, 6] - s[:, 5]).mean() * 0.01))
    # per compute unit: workgroups alive over time
    hw = stamps.view(-1, 8)[:, 7].cpu()
    hw = hw[stamps.view(-1, 8)[:, 0].cpu() > 0]
    hwid, xcc = hw & 0xffffffff, (hw >> 32) & 0xf
    cu, sh, se = (hwid >> 8) & 0xf, (hwid >> 12) & 1, (hwid >> 13) & 0x7
    key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    import collections
    per = collections.defaultdict(list)
    for k_, a_, b_ in zip(key.tolist(), s[:, 0].tolist(), s[:, 5].tolist()):
        per[k_].append((a_, b_))
    mx, avg, gaps = [], [], []
    for k_, iv in per.items():
        ev = sorted([(a_, 1) for a_, _ in iv] + [(b_, -1) for _, b_ in iv])
        cur = best = 0
        area = 0.0
        last = ev[0][0]
        for t_, d_ in ev:
            area += cur * (t_ - last)
            last = t_
            cur += d_
            best = max(best, cur)
        mx.append(best)
        avg.append(area / (ev[-1][0] - ev[0][0]))
    print("   %d compute units seen; workgroups alive per CU: max %.1f (mean over CUs), time-average %.2f; workgroups per CU %.0f" %
          (len(per), sum(mx) / len(mx), sum(avg) / len(avg), len(s) / len(per)))
    life = (s[:, 5] - s[:, 0]).mean() * 0.01
    span = (s[:, 5].max() - t0) * 0.01
    print("   workgroup lifetime %.2f us; alive on average %.0f of the %d stamped (per CU %.1f)" % (life, life * len(s) / span, len(s), life * len(s) / span / 256))
