"""GPU box: mural_snv_forward_front against the per-layer path's own second-stage input (argv[1] = distal_radius)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mural_amd import _lib  # noqa: E402
from mural_amd.data import PackedGenome  # noqa: E402
from mural_amd.model import generic_eval as G  # noqa: E402
from mural_amd.model.model_snv import POOLS_LARGE  # noqa: E402
from tests.test_gpu_snv import product_from_hp  # noqa: E402

R = int(sys.argv[1]) if len(sys.argv) > 1 else 16000
torch.manual_seed(0)
model, _ = product_from_hp(np.array([7, 3, R, 150, 75, 32, 3, 4, 2]))
for p in model.parameters():
    if p.dim() > 1:
        torch.nn.init.normal_(p, std=0.1)
model = model.cuda().eval()
rng = np.random.default_rng(1)
n = 90_000
seq = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=n).tobytes().decode()
genome = PackedGenome.from_sequence(seq, "cuda")
pos = torch.from_numpy(rng.integers(0, n, size=12)).cuda()
pos[0] = n // 2
strand = torch.zeros(12, dtype=torch.uint8).cuda()
dev = pos.device
with torch.no_grad(), torch.cuda.device(dev):
    x = genome.encode_onehot(pos, strand, R).float()
    g = lambda nm: getattr(model, nm + "_2")  # noqa: E731
    out = G._pool(G._bn_conv(x, g("conv1")[0], False, g("conv1")[1]), *POOLS_LARGE[0])
    want = G._pool(G._res_blocks(g("RBs1"), out), *POOLS_LARGE[1])            # (B, 32, L3)
    h = model._get_handle()
    lay = (C.c_int32 * 16)()
    _lib.check(_lib.lib().mural_snv_tap_layout(h, lay))
    L3 = int(lay[3])
    print("L2", lay[2], "L3", L3, "front_only", lay[10], "want", tuple(want.shape))
    s3 = torch.empty((len(pos), L3, 32), device=dev)
    ws = torch.empty(int(_lib.lib().mural_snv_workspace_bytes_min(h, len(pos), 0)), dtype=torch.uint8, device=dev)
    gs = genome.as_struct(dev)
    _lib.check(_lib.lib().mural_snv_forward_front(h, C.byref(gs), pos.data_ptr(), strand.data_ptr(), len(pos), s3.data_ptr(), ws.data_ptr(), ws.numel(),
                                                 _lib.current_stream_ptr(dev)))
    got = s3.permute(0, 2, 1)
    d = (got - want).abs()
    print("max diff", float(d.max()), "scale", float(want.abs().max()))
    col = d.amax(dim=(0, 1)).cpu().numpy()
    bad = np.nonzero(col > 1e-3)[0]
    print("bad columns", len(bad), bad[:40], "... of", L3)
    site = d.amax(dim=(1, 2)).cpu().numpy()
    print("per site", np.round(site, 4))

# the whole hybrid against the per-layer path with non-trivial BatchNorm statistics and both strands
from oracle import snv_ref, synth  # noqa: E402  (debug tool: weights only)
orc = snv_ref.build(2, local_radius=7, distal_radius=R)
sd = synth.synth_state_dict(orc.state_dict(), 5 + R)
model2, _ = product_from_hp(np.array([7, 3, R, 150, 75, 32, 3, 4, 2]))
model2.load_state_dict(sd)
model2 = model2.cuda().eval()
strand2 = torch.from_numpy(rng.integers(0, 2, size=12).astype(np.uint8)).cuda() if os.environ.get('MIXED', '1') == '1' else torch.zeros(12, dtype=torch.uint8).cuda()
with torch.no_grad(), torch.cuda.device(dev):
    from mural_amd.model.model_snv import POOLS_MID
    x = genome.encode_onehot(pos, strand2, R).float()
    cat = genome.encode_kmer(pos, strand2, 7, 3)
    L = x.shape[2]
    mid_a = G.tower(model2, "", x[:, :, L // 2 - 100:L // 2 + 101].contiguous(), POOLS_MID)
    lar_a = G.tower(model2, "_2", x.contiguous(), POOLS_LARGE)
    h = model2._get_handle()
    s3 = torch.empty((len(pos), L3, 32), device=dev)
    _lib.check(_lib.lib().mural_snv_forward_front(h, C.byref(gs), pos.data_ptr(), strand2.data_ptr(), len(pos), s3.data_ptr(), ws.data_ptr(), ws.numel(),
                                                 _lib.current_stream_ptr(dev)))
    g2 = lambda nm: getattr(model2, nm + "_2")  # noqa: E731
    out = G._pool(G._bn_conv(x, g2("conv1")[0], False, g2("conv1")[1]), *POOLS_LARGE[0])
    want = G._pool(G._res_blocks(g2("RBs1"), out), *POOLS_LARGE[1])
    d2 = (s3.permute(0, 2, 1) - want).abs()
    print("s3 diff (synth weights)", float(d2.max()), float(want.abs().max()), 'strands', strand2.tolist(), 'per site', np.round(d2.amax(dim=(1, 2)).cpu().numpy(), 3))
    colbad = np.nonzero(d2.amax(dim=(0, 1)).cpu().numpy() > 1e-3)[0]
    print('bad cols', len(colbad), colbad[:30], colbad[-10:])
    want_bn = G._bn(want, g2('conv2')[0], False)
    print('s3 vs BN_mid(pooled)', float((s3.permute(0, 2, 1) - want_bn).abs().max()), float(want_bn.abs().max()))
    lar_b = G.tower_tail(model2, "_2", s3.permute(0, 2, 1).contiguous(), POOLS_LARGE)
    mid_b = G.tower(model2, "", genome.encode_onehot(pos, strand2, 100).float(), POOLS_MID)
    print("lar diff", float((lar_a - lar_b).abs().max()), "mid diff", float((mid_a - mid_b).abs().max()))
    a = model2.forward_packed(genome, pos, strand2, local_radius=7, local_order=3)
    b = G.forward(model2, cat, x, POOLS_MID, POOLS_LARGE)
    print("hybrid vs per-layer logp", float((a - b).abs().max()))
