"""GPU-box stress: the prediction paths on a large site list, many times, bit for bit (units are handed to waves through atomic
tickets, so every run distributes the work differently -- a race or an uninitialised read would show up as a difference)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from mural_amd.data import PackedGenome  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda", 0)
codes = bench.synthetic_genome(2_200_000 + 2 * bench.DISTAL_RADIUS)
packed, mask = bench.pack2(codes)
genome = PackedGenome(packed, mask, len(codes), dev)
model = bench.build_model(dev)
n = 2_000_000
idx = torch.arange(n, device=dev, dtype=torch.int64)
pos, strand = idx + bench.DISTAL_RADIUS, (idx % 3 == 0).to(torch.uint8)
with torch.no_grad():
    ref = model.forward_packed(genome, pos, strand, local_radius=bench.LOCAL_RADIUS, local_order=bench.LOCAL_ORDER)
    ref_r = model.forward_packed_reuse(genome, pos, strand, local_radius=bench.LOCAL_RADIUS, local_order=bench.LOCAL_ORDER)
    torch.cuda.synchronize()
    assert torch.isfinite(ref).all() and torch.isfinite(ref_r).all()
    print("reuse vs per-window: max |d log p| = %.3e" % float((ref - ref_r).abs().max()))
    bad = 0
    for r in range(reps):
        a = model.forward_packed(genome, pos, strand, local_radius=bench.LOCAL_RADIUS, local_order=bench.LOCAL_ORDER)
        b = model.forward_packed_reuse(genome, pos, strand, local_radius=bench.LOCAL_RADIUS, local_order=bench.LOCAL_ORDER)
        torch.cuda.synchronize()
        ok_a, ok_b = torch.equal(a, ref), torch.equal(b, ref_r)
        bad += (not ok_a) + (not ok_b)
        if not (ok_a and ok_b):
            print("run", r, "differs: per-window", ok_a, "reuse", ok_b)
print("%d runs of 2 M sites through both paths: %s" % (reps, "bitwise identical" if bad == 0 else "%d DIFFERENCES" % bad))
sys.exit(1 if bad else 0)
