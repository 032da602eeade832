"""bench.py's SNV training leg alone (un-synchronised steps), for A/B runs: python tools/archive/r4_train_only.py [steps]"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from mural_amd.data import PackedGenome
dev = torch.device("cuda", 0)
codes = bench.synthetic_genome(4_096_000 + 2000)
packed, mask = bench.pack2(codes)
genome = PackedGenome(packed, mask, len(codes), dev)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
out = bench.train_steps_per_s(dev, genome, steps=steps)
print(json.dumps({k: v for k, v in out.items() if isinstance(v, (int, float))}))
