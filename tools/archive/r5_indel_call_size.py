import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from mural_amd.model import model_choice, weights_init
from mural_amd.data import PackedGenome
cfg = dict(CNN_out_channels=8, CNN_kernel_size=7, down_list=[1, 4, 5, 5, 5, 2], use_reverse=True)
torch.manual_seed(0)
model = model_choice(0, cfg, dict(n_class=8), "indel"); model.apply(weights_init); model = model.cuda().eval()
rng = np.random.default_rng(0)
seq = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, size=12_000_000)].tobytes().decode()
genome = PackedGenome.from_sequence(seq, "cuda")
N = 204800
idx = torch.arange(N, device="cuda", dtype=torch.int64)
pos, strand = idx * 47 + 4000, (idx & 1).to(torch.uint8)
with torch.no_grad():
    for per in (20480, 204800, 20480, 204800):
        model.forward_packed(genome, pos[:per], strand[:per], 4000); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for c0 in range(0, N, per):
            model.forward_packed(genome, pos[c0:c0 + per], strand[c0:c0 + per], 4000)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print("per call %6d: host returned after %.1f ms, device done after %.1f ms -> %.0f positions/s" % (per, (t1 - t0) * 1e3, (t2 - t0) * 1e3, N / (t2 - t0)), flush=True)
