"""GPU-box debugging aid: dump the LDS-resident stage outputs of the fused SNV kernel for tile 0 and compare each
against the CPU oracle's intermediate tensors.  Not part of the product; run via gpurun."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from mural_amd.model import model_choice  # noqa: E402
from oracle import snv_ref, synth  # noqa: E402
from tests import _util as U  # noqa: E402


def unswizzle(buf, ncols_phys):
    """physical LDS image -> [physical col][32]"""
    out = np.zeros((ncols_phys, 32), np.float32)
    for pc in range(ncols_phys):
        key = (0x2e4c11ee4587 >> (3 * (pc & 15))) & 7
        for chunk in range(8):
            src = pc * 32 + 4 * (chunk ^ key)
            out[pc, 4 * chunk:4 * chunk + 4] = buf[src:src + 4]
    return out


def stage_view(buf, P, L, nbuf):
    """-> [P][32][L] from a dumped buffer in the flattened geometry with stride L+1"""
    phys = unswizzle(buf, nbuf // 32)
    res = np.zeros((P, 32, L), np.float32)
    for p in range(P):
        for j in range(L):
            c = 1 + p * (L + 1) + j
            res[p, :, j] = phys[c + 1]
    return res


def main(r=7, R=1000, seed=21, B=6):
    torch.manual_seed(0)
    cfg = dict(local_radius=r, local_order=3, local_hidden1_size=150, local_hidden2_size=75, distal_radius=R,
               emb_dropout=0.1, local_dropout=0.1, CNN_kernel_size=3, CNN_out_channels=32, distal_fc_dropout=0.25)
    ncol = 2 * r + 1 - 2
    common = dict(emb_dims=[(65, 2)] * ncol, n_cont=0, n_class=4, distal_order=1, in_channels=4)
    orc = snv_ref.build(2, local_radius=r, distal_radius=R)
    sd = synth.synth_state_dict(orc.state_dict(), seed)
    orc.load_state_dict(sd)
    orc.eval()
    mdl = model_choice(2, cfg, common, "snv")
    mdl.load_state_dict(sd)
    mdl = mdl.cuda().eval()
    rng = np.random.default_rng(5)
    codes = rng.integers(0, 4, size=(B, 2 * R + 1)).astype(np.uint8)
    codes[1, 5:40] = 4
    codes[1, R - 100] = 4
    codes[0, 0] = 4
    cat = rng.integers(0, 65, size=(B, ncol)).astype(np.int64)
    x = U.onehot(codes)
    taps_o = {}
    with torch.no_grad():
        want = orc((torch.zeros(B, 1, dtype=torch.float64), torch.from_numpy(cat)), x, taps=taps_o).numpy()
    lay = mdl.tap_layout()
    P, nbuf = lay[0], lay[1]
    print("layout P=%d nbuf=%d large L=%s mid L=%s lds=%d" % (P, nbuf, lay[2:5], lay[5:8], lay[9]))
    taps = torch.zeros(13 * nbuf, dtype=torch.float32, device="cuda")
    with torch.no_grad():
        got = mdl((torch.zeros(B, 1, dtype=torch.float64).cuda(), torch.from_numpy(cat).cuda()), x.cuda(), _taps=taps)
    torch.cuda.synchronize()
    got = got.cpu().numpy()
    t = taps.cpu().numpy().reshape(13, nbuf)

    def bn_eval(name, v):
        m = dict(orc.named_modules())[name]
        s = (m.weight / torch.sqrt(m.running_var + m.eps)).detach().numpy()
        sh = (m.bias - m.running_mean * m.weight / torch.sqrt(m.running_var + m.eps)).detach().numpy()
        return v * s[None, :, None] + sh[None, :, None]

    for tw, sfx in ((0, "_2"), (1, "")):
        Ls = lay[2:5] if tw == 0 else lay[5:8]
        pairs = [(1, "rbs1", Ls[0], None), (2, "pool2", Ls[1], "conv2" + sfx + ".0"),
                 (3, "rbs2", Ls[1], None), (4, "pool3", Ls[2], "conv3" + sfx + ".0"), (5, "conv3", Ls[2], None)]
        for slot, name, L, bn in pairs:
            mine = stage_view(t[tw * 6 + slot], P, L, nbuf)
            ref = taps_o[name + sfx].numpy()[:P]
            if bn:
                ref = bn_eval(bn, ref)
            err = np.abs(mine - ref).max()
            print(f"tower {tw} slot {slot} {name:6s} L={L:4d} max|err|={err:.3e}  ref max={np.abs(ref).max():.3f}")
            if err > 1e-3:
                bad = np.argwhere(np.abs(mine - ref) > 1e-3)
                print("   first bad (p,ch,j):", bad[:8].tolist(), " n_bad", len(bad))
    small = t[12]
    feat = small[: 2 * P * 32].reshape(2, P, 32)
    logit = small[2 * P * 32: 2 * P * 32 + 2 * P * 16].reshape(2, P, 16)
    for tw, sfx in ((0, "_2"), (1, "")):
        fcname = "distal_fc2" if tw == 0 else "distal_fc1"
        ref_feat = taps_o["gmax" + sfx].numpy()[:P]
        print(f"tower {tw} feat err {np.abs(feat[tw] - ref_feat).max():.3e}  logits err "
              f"{np.abs(logit[tw][:, :4] - taps_o['fc' + sfx].numpy()[:P]).max():.3e}")
    print("final logp err", np.abs(got - want).max(), " prob err", np.abs(np.exp(got) - np.exp(want)).max())
    print("got[0]", got[0], "want[0]", want[0])


if __name__ == "__main__":
    main()
    main(r=10, R=1000, seed=3, B=5)
    main(r=5, R=100, seed=4, B=33)
    main(r=7, R=200, seed=6, B=17)
