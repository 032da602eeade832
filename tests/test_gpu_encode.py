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
    fx, seq, genome = enc
    codes = encode_ref.seq_to_codes(seq)
    codes_n = np.where(codes > 4, 4, codes)            # the packed format stores ambiguity codes as N
    starts = fx["starts"]
    strand = fx["strands"]
    sym = ["-" if s else "+" for s in strand]
    for R in (100, 1000):
        got = genome.encode_onehot(starts, strand, R, model_type).cpu().numpy()
        want_n = encode_ref.onehot_encode(codes_n, starts, sym, R, model_type)
        assert np.array_equal(got, want_n)
        # rows whose window holds no ambiguity code other than N equal the reference's own output
        want = encode_ref.onehot_encode(codes, starts, sym, R, model_type)
        clean = (want == want_n).all(axis=(1, 2))
        assert clean.sum() >= 3 or R == 1000
        assert np.array_equal(got[clean], want[clean])


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
