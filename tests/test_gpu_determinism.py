"""The prediction paths have no floating-point atomics and no order-dependent reductions: the same call twice gives the same bits.
(A difference would point at a race -- two lanes of a forward sharing scratch, a hand-over read before it was published.)"""
import numpy as np
import pytest
import torch

from tests import _util as U

pytestmark = pytest.mark.gpu


def _snv():
    from tests.test_gpu_snv import product_from_hp
    fx = U.load("snv_synth_S_net2.npz")
    model, _ = product_from_hp(fx["hp"])
    orc = U.snv_oracle_from_hp(fx["hp"])
    model.load_state_dict(U.snv_state_for(fx, orc))
    return model.cuda().eval(), int(fx["hp"][0]), int(fx["hp"][2])


@pytest.mark.parametrize("n_sites", [40, 3000])
def test_snv_forwards_are_bitwise_repeatable(n_sites):
    from mural_amd.data import PackedGenome
    model, r, R = _snv()
    rng = np.random.default_rng(n_sites)
    seq = rng.choice(np.frombuffer(b"ACGTNRY", np.uint8), size=40_000, p=[.245, .245, .245, .245, .01, .005, .005]).tobytes().decode()
    genome = PackedGenome.from_sequence(seq, "cuda")
    pos = torch.from_numpy(np.sort(rng.integers(0, len(seq), size=n_sites))).cuda()
    strand = torch.from_numpy(rng.integers(0, 2, size=n_sites).astype(np.uint8)).cuda()
    x = genome.encode_onehot(pos, strand, R)
    cat = genome.encode_kmer(pos, strand, r, 3)
    cont = torch.zeros(n_sites, 1, dtype=torch.float64, device="cuda")
    with torch.no_grad():
        runs = [(model((cont, cat), x), model.forward_packed(genome, pos, strand, local_radius=r, local_order=3),
                 model.forward_packed_reuse(genome, pos, strand, local_radius=r, local_order=3)) for _ in range(3)]
    for later in runs[1:]:
        for a, b in zip(runs[0], later):
            assert torch.equal(a, b)


def test_indel_forward_is_bitwise_repeatable_across_both_lanes():
    from tests.test_gpu_indel import product_from
    fx = U.load("indel_synth_small.npz")
    model = product_from(fx)
    orc = U.indel_oracle_from_hp(fx["hp"], fx["down"])
    model.load_state_dict(U.indel_state_for(fx, orc))
    model = model.cuda().eval()
    x = U.onehot(np.random.default_rng(9).integers(0, 4, size=(5000, fx["codes"].shape[1])).astype(np.uint8)).cuda()
    with torch.no_grad():
        a, b, c = model(x), model(x), model(x)
        rows = torch.cat([model(x[:2048]), model(x[2048:4096]), model(x[4096:])])      # the same chunks, one lane at a time
    assert torch.equal(a, b) and torch.equal(a, c) and torch.equal(a, rows)


def test_tower_kernel_variants_agree_bit_for_bit(monkeypatch):
    """The wave-private tower kernel hands units to waves at a fixed stride (or, behind a switch, through a counter), runs instances
    with the window geometry at compile time and runs the short stages of up to four chunks as one launch per tower: none of it may
    change a bit of the result.  Same sites with (a) the fixed stride, (b) tickets, (c) the first-stage instance that reads its
    geometry from the arguments, (d) the short-stage launches per chunk, (e) the local branch's fragments in LDS; 300 k sites = three
    chunks, several units per wave in every launch."""
    import bench
    from mural_amd.data import PackedGenome
    dev = torch.device("cuda", 0)
    codes = bench.synthetic_genome(540_000 + 2 * bench.DISTAL_RADIUS)
    packed, mask = bench.pack2(codes)
    genome = PackedGenome(packed, mask, len(codes), dev)
    model = bench.build_model(dev)
    n = 300_000
    idx = torch.arange(n, device=dev, dtype=torch.int64)
    pos, strand = idx + bench.DISTAL_RADIUS, (idx % 3 == 0).to(torch.uint8)

    def run():
        with torch.no_grad():
            out = model.forward_packed(genome, pos, strand, local_radius=bench.LOCAL_RADIUS, local_order=bench.LOCAL_ORDER)
        torch.cuda.synchronize()
        return out

    base = run()
    assert torch.isfinite(base).all()
    monkeypatch.setenv("MURAL_TOWER_DYNAMIC_UNITS", "1")      # units through the ticket counter instead of the fixed stride (the default)
    static = run()
    monkeypatch.delenv("MURAL_TOWER_DYNAMIC_UNITS")
    monkeypatch.setenv("MURAL_DEBUG_TOWER_RUNTIME_GEOM", "1")
    runtime_geom = run()
    monkeypatch.delenv("MURAL_DEBUG_TOWER_RUNTIME_GEOM")
    monkeypatch.setenv("MURAL_SNV_DEFER_SHORT", "0")      # the short-stage launches per chunk instead of once per four chunks
    per_chunk = run()
    monkeypatch.delenv("MURAL_SNV_DEFER_SHORT")
    monkeypatch.setenv("MURAL_LOCAL_REG", "0")            # the local branch's weight fragments in LDS instead of registers
    local_lds = run()
    monkeypatch.delenv("MURAL_LOCAL_REG")
    assert torch.equal(base, local_lds)
    assert torch.equal(base, static)
    assert torch.equal(base, runtime_geom)
    assert torch.equal(base, per_chunk)
    assert torch.equal(base, run())
    # more than four chunks (two short-stage launches per tower), and a last chunk of a single site
    for n2 in (4 * 131072 + 5000, 131072 + 1):
        idx = torch.arange(n2, device=dev, dtype=torch.int64)
        pos, strand = idx + bench.DISTAL_RADIUS, (idx % 3 == 0).to(torch.uint8)
        deferred = run()
        monkeypatch.setenv("MURAL_SNV_DEFER_SHORT", "0")
        per_chunk = run()
        monkeypatch.delenv("MURAL_SNV_DEFER_SHORT")
        assert torch.equal(deferred, per_chunk) and torch.equal(deferred[:131072], base[:131072])
