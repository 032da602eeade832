"""Diagnostic: phase times of the two level-0 launches of the INDEL forward on the packed entry (2048 positions of L = 8000): per workgroup
the first thread's clock at entry / front input staged / block input ready / SiLU done / block output ready / exit
(mural_debug_cb8_set_stamps; MURAL_DEBUG_CB_STAMP_ONLY picks the launch)."""
import os
os.environ.setdefault("MURAL_HIP_FLAVOR", "debug")      # validation hooks / development switches: the debug flavour of the library
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mural_amd import _lib
from mural_amd.model import model_choice, weights_init
from mural_amd.data import PackedGenome

lib = _lib.lib()
cfg = dict(CNN_out_channels=8, CNN_kernel_size=7, down_list=[1, 4, 5, 5, 5, 2], use_reverse=True)
torch.manual_seed(0)
model = model_choice(0, cfg, dict(n_class=8), "indel")
model.apply(weights_init)
model = model.cuda().eval()
B = 2048
rng = np.random.default_rng(0)
seq = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, size=2_000_000)].tobytes().decode()
genome = PackedGenome.from_sequence(seq, "cuda")
idx = torch.arange(B, device="cuda", dtype=torch.int64)
pos, strand = idx * 47 + 4000, (idx & 1).to(torch.uint8)
names = ["front input staged", "block input ready", "k=5 + SiLU", "1x1 + loads", "tail / store"]
with torch.no_grad():
    for _ in range(2):
        model.forward_packed(genome, pos, strand, 4000)
    for which in ("enc", "dec"):
        os.environ["MURAL_DEBUG_CB_STAMP_ONLY"] = which
        torch.cuda.synchronize()
        stamps = torch.zeros(8 * 65536, dtype=torch.int64, device="cuda")
        lib.mural_debug_cb8_set_stamps(stamps.data_ptr())
        model.forward_packed(genome, pos, strand, 4000)
        torch.cuda.synchronize()
        lib.mural_debug_cb8_set_stamps(None)
        s = stamps.view(-1, 8).cpu().double()
        s = s[s[:, 0] > 0]
        if which == "enc" and os.environ.get("MURAL_INDEL_ENC0", "1") != "0":
            # the persistent kernel: per-workgroup phase SUMS (word 7: tiles)
            tiles = s[:, 7].clamp(min=1)
            print("enc (persistent): %d workgroups, %.1f tiles each; start-up %.2f us per workgroup" % (len(s), tiles.mean(), s[:, 0].mean() * 0.01))
            for k, n in ((1, "symbols -> planes"), (2, "front"), (3, "k=5 + SiLU"), (4, "1x1 + store")):
                print("   %-20s %.2f us per tile" % (n, (s[:, k] / tiles).mean() * 0.01))
            print("   sum %.2f us per tile" % ((s[:, 1:5].sum(1) / tiles).mean() * 0.01))
            tot = s[:, 0:5].sum(1) * 0.01
            start = (s[:, 6] - s[:, 6].min()) * 0.01
            print("   per workgroup: total min %.1f / median %.1f / max %.1f us; starts min 0 / median %.1f / max %.1f us; launch span %.1f us" %
                  (tot.min(), tot.median(), tot.max(), start.median(), start.max(), (start + tot).max()))
            continue
        if which == "dec" and os.environ.get("MURAL_INDEL_DEC0", "1") != "0":
            tiles = s[:, 7].clamp(min=1)
            print("dec (persistent): %d workgroups, %.1f tiles each; start-up %.2f us per workgroup" % (len(s), tiles.mean(), s[:, 0].mean() * 0.01))
            for k, n in ((1, "source -> LDS"), (2, "polyphase front"), (3, "k=5 + SiLU"), (4, "1x1 + skip"), (5, "tail / store")):
                print("   %-20s %.2f us per tile" % (n, (s[:, k] / tiles).mean() * 0.01))
            print("   sum %.2f us per tile" % ((s[:, 1:6].sum(1) / tiles).mean() * 0.01))
            tot = s[:, 0:6].sum(1) * 0.01
            start = (s[:, 6] - s[:, 6].min()) * 0.01
            print("   per workgroup: total min %.1f / median %.1f / max %.1f us; starts min 0 / median %.1f / max %.1f us; launch span %.1f us" %
                  (tot.min(), tot.median(), tot.max(), start.median(), start.max(), (start + tot).max()))
            continue
        print(which, ": %d workgroups stamped, launch span %.1f us, workgroup lifetime %.2f us" %
              (len(s), (s[:, 5].max() - s[:, 0].min()) * 0.01, (s[:, 5] - s[:, 0]).mean() * 0.01))
        for n, (a, b) in zip(names, [(0, 1), (1, 2), (2, 3), (3, 4), (4, 5)]):
            print("   %-20s %.2f us" % (n, (s[:, b] - s[:, a]).mean() * 0.01))
