"""GPU parity of the cross-position reuse path (csrc/snv_reuse.hip): for dense same-strand site lists it
must return what the per-window path returns (which the other GPU tests hold to the reference's goldens / the oracle) within
1e-5 on probabilities -- both strands, chromosome ends, N runs, IUPAC codes, several window radii, chunked spans."""
import numpy as np
import pytest
import torch

from oracle import encode_ref, snv_ref, synth
from tests import _util as U

pytestmark = pytest.mark.gpu

PROB_TOL = 1e-5


def _pair(r, R, model_no=2, seed=301, n_class=4):
    from tests.test_gpu_snv import product_from_hp
    orc = snv_ref.build(model_no, local_radius=r, distal_radius=R, n_class=n_class)
    sd = synth.synth_state_dict(orc.state_dict(), seed)
    orc.load_state_dict(sd)
    orc.eval()
    model, _ = product_from_hp(np.array([r, 3, R, 150, 75, 32, 3, n_class, model_no]))
    model.load_state_dict(sd)
    return model.cuda().eval(), orc


def _genome(rng, n, iupac=True):
    alphabet = b"ACGTNRYSWB" if iupac else b"ACGTN"
    p = [.2465, .2465, .2465, .2465, .009, .001, .001, .001, .001, .001] if iupac else [.2475, .2475, .2475, .2475, .01]
    raw = rng.choice(np.frombuffer(alphabet, np.uint8), size=n, p=p)
    raw[n // 3:n // 3 + 70] = ord("N")
    return raw.tobytes().decode()


def _prob(logp):
    return np.exp(logp.astype(np.float64))


@pytest.mark.parametrize("cfg", [(10, 1000, 2), (7, 1000, 1), (7, 300, 2), (5, 128, 2), (7, 200, 2)])
def test_reuse_equals_per_window_path_and_oracle(cfg):
    from mural_amd.data import PackedGenome
    r, R, model_no = cfg
    model, orc = _pair(r, R, model_no)
    rng = np.random.default_rng(500 + R)
    n = 30_000
    seq = _genome(rng, n)
    genome = PackedGenome.from_sequence(seq, "cuda")
    # dense runs on both strands, both chromosome ends, a sparse tail
    pos = np.r_[np.arange(0, 700), np.arange(9000, 12500), np.arange(n - 650, n), rng.integers(0, n, size=300)]
    strand = (np.arange(len(pos)) % 2).astype(np.uint8)
    strand[700:1400] = 1
    perm = rng.permutation(len(pos))                    # the entry takes any order and both strands in one call
    pos, strand = pos[perm], strand[perm]
    tp, ts = torch.from_numpy(pos).cuda(), torch.from_numpy(strand).cuda()
    with torch.no_grad():
        want = model.forward_packed(genome, tp, ts, local_radius=r, local_order=3).cpu().numpy()
        got = model.forward_packed_reuse(genome, tp, ts, local_radius=r, local_order=3).cpu().numpy()
    assert np.isfinite(got).all()
    err = np.abs(_prob(got) - _prob(want)).max()
    assert err <= PROB_TOL, f"{cfg}: reuse vs per-window {err:.3e}"
    sel = rng.choice(len(pos), size=96, replace=False)
    codes = encode_ref.seq_to_codes(seq)
    sym = ["-" if s else "+" for s in strand[sel]]
    with torch.no_grad():
        ref = orc((torch.zeros(len(sel), 1, dtype=torch.float64), torch.from_numpy(encode_ref.kmer_encode(codes, pos[sel], sym, r, 3))),
                  torch.from_numpy(encode_ref.onehot_encode(codes, pos[sel], sym, R))).numpy()
    assert np.abs(_prob(got[sel]) - _prob(ref)).max() <= PROB_TOL, f"{cfg}: reuse vs oracle"


def test_reuse_one_million_dense_sites_both_strands():
    """a 1 M-site dense list, both strands, reuse == per-window within 1e-5."""
    from mural_amd.data import PackedGenome
    model, _ = _pair(10, 1000)
    rng = np.random.default_rng(77)
    n = 1_010_000
    genome = PackedGenome.from_sequence(_genome(rng, n, iupac=False), "cuda")
    pos = torch.arange(2000, 2000 + 1_000_000, device="cuda")
    strand = (pos & 1).to(torch.uint8)
    with torch.no_grad():
        want = model.forward_packed(genome, pos, strand, local_radius=10, local_order=3)
        got = model.forward_packed_reuse(genome, pos, strand, local_radius=10, local_order=3)
    err = float((got.double().exp() - want.double().exp()).abs().max())
    assert err <= PROB_TOL, err
    assert torch.isfinite(got).all()


def test_reuse_chunks_long_spans_and_falls_back():
    from mural_amd import _lib
    from mural_amd.data import PackedGenome
    model, _ = _pair(7, 1000, seed=302)
    span = int(_lib.lib().mural_snv_reuse_chunk_span())
    rng = np.random.default_rng(78)
    n = 2 * span + 300_000
    codes = rng.integers(0, 4, size=n).astype(np.uint8)
    genome = PackedGenome.from_sequence(np.frombuffer(b"ACGT", np.uint8)[codes].tobytes().decode(), "cuda")
    pos = np.r_[np.arange(5000, 9000), np.arange(span - 2000, span + 2000), np.arange(2 * span + 100_000, 2 * span + 104_000)]
    strand = (rng.integers(0, 2, size=len(pos))).astype(np.uint8)
    tp, ts = torch.from_numpy(pos).cuda(), torch.from_numpy(strand).cuda()
    with torch.no_grad():
        want = model.forward_packed(genome, tp, ts, local_radius=7, local_order=3)
        got = model.forward_packed_reuse(genome, tp, ts, local_radius=7, local_order=3)
    assert float((got.double().exp() - want.double().exp()).abs().max()) <= PROB_TOL
    # a window too short for the edge pyramids (R = 100: 14 pooled columns) and the local-only model take the per-window path
    small, _ = _pair(5, 100, seed=303)
    assert not _lib.lib().mural_snv_reuse_supported(small._get_handle())
    with torch.no_grad():
        a = small.forward_packed(genome, tp[:500], ts[:500], local_radius=5, local_order=3)
        b = small.forward_packed_reuse(genome, tp[:500], ts[:500], local_radius=5, local_order=3)
    assert torch.equal(a, b)
    assert model.forward_packed_reuse(genome, tp[:0], ts[:0], local_radius=7, local_order=3).shape == (0, 4)


def test_reuse_does_not_read_unwritten_workspace(monkeypatch):
    """The reuse path's workspace (per-base rows, pooled tiles, edge columns, site maps) with 0xFF-poisoned allocations: same result."""
    from mural_amd.data import PackedGenome
    from mural_amd.model import model_snv as MS
    from tests.test_gpu_snv import _PoisonedTorch
    model, _ = _pair(10, 1000, 2)
    rng = np.random.default_rng(77)
    n = 30_000
    genome = PackedGenome.from_sequence(_genome(rng, n), "cuda")
    pos = np.r_[np.arange(0, 900), np.arange(12000, 15000), np.arange(n - 500, n), rng.integers(0, n, size=200)]
    strand = (np.arange(len(pos)) % 2).astype(np.uint8)
    tp, ts = torch.from_numpy(pos).cuda(), torch.from_numpy(strand).cuda()
    with torch.no_grad():
        want = model.forward_packed_reuse(genome, tp, ts, local_radius=10, local_order=3).cpu().numpy()
        monkeypatch.setattr(MS, "torch", _PoisonedTorch())
        model._ws = None
        model._ws_rows = [0, 0]
        got = model.forward_packed_reuse(genome, tp, ts, local_radius=10, local_order=3).cpu().numpy()
    assert np.isfinite(got).all()
    assert np.abs(got - want).max() <= 2e-6


def test_reuse_writes_stay_inside_their_workspace_regions(monkeypatch):
    """The reuse path's workspace with 4 KB of poisoned guard bytes behind every region (per-base rows of both strands, k-mer ids,
    logits, hand-over tiles): every guard byte survives the call."""
    import ctypes as C
    from mural_amd import _lib
    from mural_amd.data import PackedGenome
    from mural_amd.model import model_snv as MS
    from tests.test_gpu_snv import _PoisonedTorch
    guard = 4096
    monkeypatch.setenv("MURAL_DEBUG_WS_GUARD", str(guard))
    monkeypatch.setattr(MS, "torch", _PoisonedTorch())
    model, _ = _pair(10, 1000, 2)
    rng = np.random.default_rng(78)
    n = 30_000
    genome = PackedGenome.from_sequence(_genome(rng, n), "cuda")
    pos = np.r_[np.arange(0, 900), np.arange(12000, 15000), np.arange(n - 500, n), rng.integers(0, n, size=200)]
    strand = (np.arange(len(pos)) % 2).astype(np.uint8)
    with torch.no_grad():
        out = model.forward_packed_reuse(genome, torch.from_numpy(pos).cuda(), torch.from_numpy(strand).cuda(), local_radius=10, local_order=3)
    assert torch.isfinite(out).all()
    layout = (C.c_size_t * 128)()
    n_regions = _lib.lib().mural_debug_last_ws_layout(layout, 64)
    assert n_regions >= 20
    ws = model._ws.cpu().numpy()
    for i in range(n_regions):
        off, size = layout[2 * i], layout[2 * i + 1]
        zone = ws[off + size:off + size + guard]
        assert len(zone) == guard and (zone == 255).all(), f"region {i}: a kernel wrote behind its {size} bytes"


def test_reuse_span_multiple_keeps_last_site_and_density_fallback():
    """with p_max - p_min an exact multiple of the chunk span the site at p_max fell outside every half-open chunk and
    its row stayed uninitialised.  Also: chunks below `min_density` take the per-window kernels, rows keep their order."""
    from mural_amd import _lib
    from mural_amd.data import PackedGenome
    model, _ = _pair(7, 1000, seed=304)
    span = int(_lib.lib().mural_snv_reuse_chunk_span())
    rng = np.random.default_rng(79)
    n = 2 * span + 10_000
    codes = rng.integers(0, 4, size=n).astype(np.uint8)
    genome = PackedGenome.from_sequence(np.frombuffer(b"ACGT", np.uint8)[codes].tobytes().decode(), "cuda")
    p0 = 3000
    pos = np.r_[np.arange(p0, p0 + 3000), np.arange(p0 + span - 1500, p0 + span + 1500), np.arange(p0 + 2 * span - 2000, p0 + 2 * span + 1)]
    assert (pos.max() - pos.min()) % span == 0
    strand = (rng.integers(0, 2, size=len(pos))).astype(np.uint8)
    perm = rng.permutation(len(pos))
    tp, ts = torch.from_numpy(pos[perm]).cuda(), torch.from_numpy(strand[perm]).cuda()
    with torch.no_grad():
        want = model.forward_packed(genome, tp, ts, local_radius=7, local_order=3)
        got, used = model.forward_packed_reuse(genome, tp, ts, local_radius=7, local_order=3, return_reuse_count=True)
        assert used == len(pos)
        assert torch.isfinite(got).all()
        assert float((got.double().exp() - want.double().exp()).abs().max()) <= PROB_TOL
        # every chunk holds ~3000 sites on a 2 M-base span: below any sensible density -> per-window kernels, bitwise the same rows
        sparse, used = model.forward_packed_reuse(genome, tp, ts, local_radius=7, local_order=3, min_density=0.1, batch_sites=1000,
                                                  return_reuse_count=True)
        assert used == 0 and torch.equal(sparse, want)
        # single-chunk form of the same rule
        few = model.forward_packed_reuse(genome, tp[:50], ts[:50], local_radius=7, local_order=3, min_density=0.1)
        assert torch.equal(few, model.forward_packed(genome, tp[:50], ts[:50], local_radius=7, local_order=3))


@pytest.mark.parametrize("R", [2000, 4000])
def test_reuse_entry_on_long_windows_falls_back_to_per_window(R):
    """long-window models (segmented first stage, MuralSnvModel::longwin) set `split`, so the reuse entry's
    support test used to say yes and the launch then refused the pooled tile ('reuse: pooled tile too large') on every dense
    shard.  mural_snv_reuse_supported now evaluates every geometry condition the launches require: the Python entry must fall
    back to the per-window kernels (used == 0) with identical rows, and the C entry must refuse up front."""
    import ctypes as C
    from mural_amd import _lib
    from mural_amd.data import PackedGenome
    model, _ = _pair(7, R, seed=310 + R)
    assert model._fused_ok()
    lib = _lib.lib()
    assert not lib.mural_snv_reuse_supported(model._get_handle())
    rng = np.random.default_rng(R)
    n = 12_000
    genome = PackedGenome.from_sequence(_genome(rng, n, iupac=False), "cuda")
    pos = torch.arange(100, 100 + 1500, device="cuda")                # dense: 1500 sites on 1500 bases, >= REUSE_MIN_SITES
    strand = (pos & 1).to(torch.uint8)
    with torch.no_grad():
        want = model.forward_packed(genome, pos, strand, local_radius=7, local_order=3)
        got, used = model.forward_packed_reuse(genome, pos, strand, local_radius=7, local_order=3, min_density=0.1,
                                               return_reuse_count=True)
    assert used == 0 and torch.equal(got, want)
    g = genome.as_struct(pos.device)
    out = torch.empty((pos.shape[0], 4), device="cuda")
    ws = torch.empty(1 << 20, dtype=torch.uint8, device="cuda")
    rc = lib.mural_snv_forward_packed_reuse(model._get_handle(), C.byref(g), pos.data_ptr(), strand.data_ptr(), pos.shape[0], 3, 100, 1599,
                                            7, 3, out.data_ptr(), ws.data_ptr(), ws.numel(), _lib.current_stream_ptr(pos.device))
    assert rc != 0 and b"reuse" in lib.mural_last_error()
