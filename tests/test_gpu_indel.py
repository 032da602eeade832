"""GPU parity: HIP UNet_Small forward vs the reference's golden vectors (relative tolerance 1e-4 on the positive
Softplus scores, 1e-5 abs on the softmax probabilities callers derive from them)."""
import glob
import os

import numpy as np
import pytest
import torch

from tests import _util as U

pytestmark = pytest.mark.gpu
INDEL_FORWARD = sorted(os.path.basename(p) for p in glob.glob(os.path.join(U.GOLDEN, "indel_*.npz")))


def product_from(fx):
    from mural_amd.model import model_choice
    R, C, k, n_class, rev = [int(v) for v in fx["hp"]]
    cfg = dict(CNN_out_channels=C, CNN_kernel_size=k, down_list=[int(d) for d in fx["down"]], use_reverse=bool(rev))
    return model_choice(0, cfg, dict(n_class=n_class), "indel")


@pytest.mark.parametrize("name", INDEL_FORWARD)
def test_forward_matches_reference(name):
    fx = U.load(name)
    model = product_from(fx)
    orc = U.indel_oracle_from_hp(fx["hp"], fx["down"])
    assert list(model.state_dict().keys()) == list(orc.state_dict().keys())
    model.load_state_dict(U.indel_state_for(fx, orc), strict=True)
    model = model.cuda().eval()
    with torch.no_grad():
        out = model(U.onehot(fx["codes"]).cuda()).cpu().numpy()
    want = fx["out"]
    assert out.shape == want.shape
    assert np.abs(out - want).max() <= 1e-4 * max(1.0, np.abs(want).max()), name
    sm = lambda a: np.exp(a - a.max(1, keepdims=True)) / np.exp(a - a.max(1, keepdims=True)).sum(1, keepdims=True)
    assert np.abs(sm(out.astype(np.float64)) - sm(want.astype(np.float64))).max() <= 1e-5


def test_batch_larger_than_chunk_and_packed_path():
    from mural_amd.data import PackedGenome
    from oracle import encode_ref
    fx = U.load("indel_synth_small.npz")
    model = product_from(fx)
    orc = U.indel_oracle_from_hp(fx["hp"], fx["down"])
    sd = U.indel_state_for(fx, orc)
    model.load_state_dict(sd)
    orc.load_state_dict(sd)
    model = model.cuda().eval()
    orc.eval()
    rng = np.random.default_rng(3)
    seq = rng.choice(np.frombuffer(b"ACGTN", np.uint8), size=30000, p=[.247, .247, .247, .247, .012]).tobytes().decode()
    codes = encode_ref.seq_to_codes(seq)
    pos = rng.integers(0, len(seq), size=300)
    pos[:3] = [0, 2, len(seq) - 1]
    strand = rng.integers(0, 2, size=300).astype(np.uint8)
    sym = ["-" if s else "+" for s in strand]
    R = int(fx["hp"][0])
    x = torch.from_numpy(encode_ref.onehot_encode(codes, pos, sym, R, "indel"))
    with torch.no_grad():
        want = orc(x).numpy()
    genome = PackedGenome.from_sequence(seq, "cuda")
    got = model.forward_packed(genome, torch.from_numpy(pos).cuda(), torch.from_numpy(strand).cuda(), R).cpu().numpy()
    assert np.abs(got - want).max() <= 1e-4 * max(1.0, np.abs(want).max())


def test_incompatible_length_is_rejected():
    fx = U.load("indel_synth_small.npz")
    model = product_from(fx).cuda().eval()
    with pytest.raises(ValueError):
        model(torch.zeros(2, 4, 800, device="cuda"))
