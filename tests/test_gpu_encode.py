"""GPU parity: packed-genome window encoders vs the golden vectors / oracle (bit exact)."""
import numpy as np
import pytest
import torch

from oracle import encode_ref
from tests import _util as U

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def enc():
    from mural_amd.data import PackedGenome
    fx = U.load("encode.npz")
    seq = fx["seq"].tobytes().decode()
    return fx, seq, PackedGenome.from_sequence(seq, "cuda")


@pytest.mark.parametrize("model_type", ["snv", "indel"])
def test_kmer_bit_exact_vs_reference_vectors(enc, model_type):
    fx, seq, genome = enc
    for neg in (False, True):
        sel = fx["strands"].astype(bool) == neg
        starts = fx["starts"][sel]
        strand = np.full(len(starts), int(neg), np.uint8)
        tag = "neg" if neg else "pos"
        for r, k in [(5, 3), (7, 3), (10, 3), (7, 1), (6, 2)]:
            got = genome.encode_kmer(starts, strand, r, k, model_type).cpu().numpy()
            want = fx[f"kmer_{model_type}_{tag}_r{r}_k{k}"]
            assert got.dtype == np.int64 and got.shape == want.shape
            assert np.array_equal(got, want), (model_type, tag, r, k)


@pytest.mark.parametrize("model_type", ["snv", "indel"])
def test_onehot_exact(enc, model_type):
    """Every row equals the reference's seq_ohe_encoder output, including the windows that hold the ten IUPAC codes other than
    N of the G1 string (fractional columns, preprocessing.py:762-772; complemented on the '-' strand)."""
    fx, seq, genome = enc
    assert len(genome.ambiguous[0]) == 10
    codes = encode_ref.seq_to_codes(seq)
    for neg in (False, True):
        sel = fx["strands"].astype(bool) == neg
        starts = fx["starts"][sel]
        strand = np.full(len(starts), int(neg), np.uint8)
        tag = "neg" if neg else "pos"
        for R in (100, 1000):
            got = genome.encode_onehot(starts, strand, R, model_type).cpu().numpy()
            assert list(got.shape) == fx[f"ohesum_{model_type}_{tag}_R{R}"].tolist()
            w = (np.arange(got.shape[2], dtype=np.float64) % 97 + 1.0)
            assert np.array_equal((got.astype(np.float64) * w[None, None, :]).sum(axis=2), fx[f"ohechk_{model_type}_{tag}_R{R}"])
            if R == 100:
                want = fx[f"ohe_{model_type}_{tag}_R{R}"]
                assert np.array_equal(got, want)
                frac = ((want != 0) & (want != 1) & (want != 0.25)).any(axis=(1, 2))
                assert frac.sum() >= 3           # the comparison does cover IUPAC windows
            assert np.array_equal(got, encode_ref.onehot_encode(codes, starts, ["-" if neg else "+"] * len(starts), R, model_type))


def test_iupac_side_table_dense_genome():
    """Many ambiguity codes (beyond one 64-entry probe round of the wave search), both strands, chromosome edges."""
    from mural_amd.data import PackedGenome
    rng = np.random.default_rng(11)
    n = 40_000
    raw = rng.choice(np.frombuffer(b"ACGTNRYMSWKBDHVacgtnryk", np.uint8), size=n,
                     p=[.2, .2, .2, .2, .02] + [.01] * 10 + [.02] * 4 + [.0] * 4)
    seq = raw.tobytes().decode()
    genome = PackedGenome.from_sequence(seq, "cuda")
    assert len(genome.ambiguous[0]) > 3000
    codes = encode_ref.seq_to_codes(seq)
    pos = np.r_[rng.integers(0, n, size=200), [0, 1, n - 1, n - 2]]
    strand = rng.integers(0, 2, size=len(pos)).astype(np.uint8)
    sym = ["-" if s else "+" for s in strand]
    for R, mt in ((100, "snv"), (1000, "snv"), (300, "indel")):
        got = genome.encode_onehot(pos, strand, R, mt).cpu().numpy()
        assert np.array_equal(got, encode_ref.onehot_encode(codes, pos, sym, R, mt))
    assert np.array_equal(genome.encode_kmer(pos, strand, 7, 3).cpu().numpy(), encode_ref.kmer_encode(codes, pos, sym, 7, 3))


def test_large_random_genome_roundtrip():
    from mural_amd.data import PackedGenome
    rng = np.random.default_rng(7)
    n = 300_000
    raw = rng.choice(np.frombuffer(b"ACGTN", np.uint8), size=n, p=[.245, .245, .245, .245, .02])
    seq = raw.tobytes().decode()
    genome = PackedGenome.from_sequence(seq, "cuda")
    codes = encode_ref.seq_to_codes(seq)
    pos = rng.integers(-5, n + 5, size=4000)
    pos = np.clip(pos, 0, n - 1)
    strand = rng.integers(0, 2, size=len(pos)).astype(np.uint8)
    sym = ["-" if s else "+" for s in strand]
    got = genome.encode_kmer(pos, strand, 10, 3).cpu().numpy()
    assert np.array_equal(got, encode_ref.kmer_encode(codes, pos, sym, 10, 3))
    got = genome.encode_onehot(pos[:300], strand[:300], 1000).cpu().numpy()
    assert np.array_equal(got, encode_ref.onehot_encode(codes, pos[:300], sym[:300], 1000))
    # size-independent property: one-hot columns sum to 1, reverse strand = flipped forward strand
    fwd = genome.encode_onehot(pos[:300], np.zeros(300, np.uint8), 1000)
    rev = genome.encode_onehot(pos[:300], np.ones(300, np.uint8), 1000)
    assert torch.equal(rev, fwd.flip([1, 2]))
    assert torch.all(fwd.sum(dim=1) == 1.0)


def test_empty_batch(enc):
    fx, seq, genome = enc
    out = genome.encode_kmer(np.zeros(0, np.int64), np.zeros(0, np.uint8), 7, 3)
    assert out.shape == (0, 13)
    out = genome.encode_onehot(np.zeros(0, np.int64), np.zeros(0, np.uint8), 100)
    assert out.shape == (0, 4, 201)
