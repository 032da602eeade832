"""GPU box: stress of the latency-shaped launches (calls of up to 256 sites: two workgroups per site hand their logits over through
global memory and an arrival counter).  Thousands of calls of random sizes, each compared with the throughput-shaped launches on the
same sites; calls are issued back to back so that hand-overs of consecutive calls overlap on the device."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from mural_amd.data import PackedGenome  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
dev = torch.device("cuda", 0)
model = bench.build_model(dev)
codes = bench.synthetic_genome(2_000_000)
packed, mask = bench.pack2(codes)
genome = PackedGenome(packed, mask, len(codes), dev)
rng = np.random.default_rng(0)
N = 256 * 64
pos = torch.from_numpy(rng.integers(0, len(codes), size=N)).to(dev)
strand = torch.from_numpy(rng.integers(0, 2, size=N).astype(np.uint8)).to(dev)
with torch.no_grad():
    os.environ["MURAL_DEBUG_NO_SMALL_BATCH"] = "1"
    ref = model.forward_packed(genome, pos, strand, local_radius=10, local_order=3)
    x = genome.encode_onehot(pos, strand, 1000)
    cat = genome.encode_kmer(pos, strand, 10, 3)
    ref_d = model((torch.zeros(N, 1, device=dev), cat), x)
    del os.environ["MURAL_DEBUG_NO_SMALL_BATCH"]
    assert float((ref - ref_d).abs().max()) <= 2e-6
    worst, outs = 0.0, []
    for it in range(iters):
        n = int(rng.integers(1, 257))
        o = int(rng.integers(0, N - n))
        if it % 2:
            got = model.forward_packed(genome, pos[o:o + n], strand[o:o + n], local_radius=10, local_order=3)
        else:
            got = model((torch.zeros(n, 1, device=dev), cat[o:o + n]), x[o:o + n])
        outs.append((o, n, got))
        if len(outs) == 200:                      # compare in batches: no sync between the calls themselves
            for oo, nn_, g in outs:
                d = float((g - ref[oo:oo + nn_]).abs().max())
                assert torch.isfinite(g).all() and d <= 2e-6, (it, oo, nn_, d)
                worst = max(worst, d)
            outs.clear()
print("stress ok: %d calls, worst difference %.2e" % (iters, worst))
