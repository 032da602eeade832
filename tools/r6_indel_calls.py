import json, os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from mural_amd.data import PackedGenome
dev = torch.device("cuda", 0)
codes = bench.synthetic_genome(bench.GENOME_SITES + 2 * bench.DISTAL_RADIUS)
packed, mask = bench.pack2(codes)
genome = PackedGenome(packed, mask, len(codes), dev)
for n, chunk in ((196608, 24576), (196608, 49152), (196608, 98304), (196608, 24576)):
    r = bench.indel_positions_per_s(dev, genome, n=n, chunk=chunk)
    print(chunk, round(r["positions_per_s"]))
