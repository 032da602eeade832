import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from tests.test_gpu_snv import snv_ref, synth, product_from_hp, encode_ref
from mural_amd.data import PackedGenome
from mural_amd.model import model_snv as MS, generic_eval
r, R, n_sites = 7, int(sys.argv[1]) if len(sys.argv) > 1 else 2000, int(sys.argv[2]) if len(sys.argv) > 2 else 300
orc = snv_ref.build(2, local_radius=r, distal_radius=R)
sd = synth.synth_state_dict(orc.state_dict(), 11)
model, _ = product_from_hp(np.array([r, 3, R, 150, 75, 32, 3, 4, 2]))
model.load_state_dict(sd)
model = model.cuda().eval()
rng = np.random.default_rng(3)
seq = rng.choice(np.frombuffer(b"ACGTN", np.uint8), size=30_000, p=[.247, .247, .247, .247, .012]).tobytes().decode()
codes = encode_ref.seq_to_codes(seq)
pos = rng.integers(0, len(seq), size=n_sites)
strand = rng.integers(0, 2, size=n_sites).astype(np.uint8)
sym = ["-" if s else "+" for s in strand]
cat = torch.from_numpy(encode_ref.kmer_encode(codes, pos, sym, r, 3)).cuda()
x = torch.from_numpy(encode_ref.onehot_encode(codes, pos, sym, R)).cuda()
cont = torch.zeros(n_sites, 1, dtype=torch.float64, device="cuda")
with torch.no_grad():
    want = generic_eval.forward(model, cat, x, MS.POOLS_MID, MS.POOLS_LARGE).cpu().numpy()
    got = model((cont, cat), x).cpu().numpy()
    orc.load_state_dict(sd); orc.eval()
    ref = orc((torch.zeros(n_sites, 1, dtype=torch.float64), cat.cpu()), x.cpu()).numpy()
d = np.abs(got - want).max(1)
print("fused", model._fused_ok(), "max diff fused-vs-perlayer", d.max(), "sites over 2e-5:", np.nonzero(d > 2e-5)[0][:20], "of", n_sites)
print("fused vs oracle", np.abs(got - ref).max(), " per-layer vs oracle", np.abs(want - ref).max())
