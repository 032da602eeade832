"""GPU box: the training leg of bench.py after another leg of it (argv[1]: config5 | indel | predict_m | none), product flavour unless
MURAL_HIP_FLAVOR says otherwise -- to find what in a full bench run disturbs the training leg."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from mural_amd.data import PackedGenome  # noqa: E402

dev = torch.device("cuda", 0)
codes = bench.synthetic_genome(4_096_000 + 2000)
packed, mask = bench.pack2(codes)
genome = PackedGenome(packed, mask, len(codes), dev)
which = sys.argv[1] if len(sys.argv) > 1 else "none"
if which.startswith("config5"):
    if "noalign" in which:
        import mural_amd.predict as P
        P._ALIGNED_BLOCKS = False
    if "noshare" in which:
        bench.rank_share = lambda *a, **k: {"seconds": 1.0}
    r = bench.config5_e2e(dev)
    print("config5", round(r["rows_per_s"]))
elif which == "indel":
    r = bench.indel_positions_per_s(dev, genome)
    print("indel", round(r["positions_per_s"]))
out = bench.train_steps_per_s(dev, genome, steps=400, sync_steps=30)
print(which, json.dumps({k: round(v, 2) for k, v in out.items() if k.startswith("steps_per_s")}))
