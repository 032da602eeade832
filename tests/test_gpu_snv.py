"""GPU parity: HIP SNV forward (through the C ABI) vs the reference's golden vectors and the CPU oracle.

Tolerance (BASELINE.json north_star): per-class probabilities within 1e-5 abs of the reference CPU fp32 path.
"""
import glob
import os

import numpy as np
import pytest
import torch
import torch.nn as nn

from oracle import encode_ref, snv_ref, synth
from tests import _util as U

pytestmark = pytest.mark.gpu

PROB_TOL = 1e-5
SNV_FORWARD = sorted(os.path.basename(p) for p in glob.glob(os.path.join(U.GOLDEN, "snv_synth_*.npz"))
                     + glob.glob(os.path.join(U.GOLDEN, "snv_pretrained_*.npz")))


def product_from_hp(hp):
    from mural_amd.model import model_choice
    r, order, R, h1, h2, C, k, n_class = [int(v) for v in hp[:8]]
    model_no = int(hp[8]) if len(hp) > 8 else 2
    ncol = 2 * r + 1 - (order - 1)
    cfg = dict(local_radius=r, local_order=order, local_hidden1_size=h1, local_hidden2_size=h2, distal_radius=R,
               emb_dropout=0.1, local_dropout=0.1, CNN_kernel_size=k, CNN_out_channels=C, distal_fc_dropout=0.25)
    common = dict(emb_dims=[(4 ** order + 1, 2)] * ncol, n_cont=0, n_class=n_class, distal_order=1, in_channels=4)
    return model_choice(model_no, cfg, common, "snv"), model_no


def assert_probs_close(got_logp, want_logp, model_no, name=""):
    if model_no == 0:   # raw logits
        assert np.abs(got_logp - want_logp).max() <= 2e-5 * max(1.0, np.abs(want_logp).max()), name
        return
    err = np.abs(np.exp(got_logp.astype(np.float64)) - np.exp(want_logp.astype(np.float64))).max()
    assert err <= PROB_TOL, f"{name}: max prob err {err:.3e}"
    big = want_logp > -12
    assert np.abs(got_logp - want_logp)[big].max() <= 2e-3, name


@pytest.mark.parametrize("name", SNV_FORWARD)
def test_forward_dense_matches_reference(name):
    fx = U.load(name)
    model, model_no = product_from_hp(fx["hp"])
    orc = U.snv_oracle_from_hp(fx["hp"])
    model.load_state_dict(U.snv_state_for(fx, orc))
    model = model.cuda().eval()
    cat = torch.from_numpy(fx["cat"]).cuda()
    x = U.onehot(fx["codes"]).cuda()
    cont = torch.zeros(len(cat), 1, dtype=torch.float64, device="cuda")
    with torch.no_grad():
        out = model((cont, cat), x)
    assert out.shape == (len(cat), int(fx["hp"][7])) and out.dtype == torch.float32 and out.is_cuda
    want = fx["logp"] if "logp" in fx.files else fx["out"]
    assert_probs_close(out.cpu().numpy(), want, model_no, name)


@pytest.mark.parametrize("batch", [1, 2, 3, 7, 64, 257])
def test_ragged_batches(batch):
    fx = U.load("snv_synth_T_net2.npz")
    model, _ = product_from_hp(fx["hp"])
    orc = U.snv_oracle_from_hp(fx["hp"])
    sd = U.snv_state_for(fx, orc)
    model.load_state_dict(sd)
    orc.load_state_dict(sd)
    model = model.cuda().eval()
    orc.eval()
    rng = np.random.default_rng(batch)
    codes = rng.integers(0, 4, size=(batch, 201)).astype(np.uint8)
    cat = rng.integers(0, 65, size=(batch, 9)).astype(np.int64)
    with torch.no_grad():
        want = orc((torch.zeros(batch, 1, dtype=torch.float64), torch.from_numpy(cat)), U.onehot(codes)).numpy()
        got = model((torch.zeros(batch, 1, dtype=torch.float64).cuda(), torch.from_numpy(cat).cuda()),
                    U.onehot(codes).cuda()).cpu().numpy()
    assert_probs_close(got, want, 2, f"batch {batch}")


def test_empty_batch():
    fx = U.load("snv_synth_T_net2.npz")
    model, _ = product_from_hp(fx["hp"])
    model = model.cuda().eval()
    out = model((torch.zeros(0, 1).cuda(), torch.zeros(0, 9, dtype=torch.long).cuda()), torch.zeros(0, 4, 201).cuda())
    assert out.shape == (0, 4)


@pytest.mark.parametrize("cfg", [(10, 1000, 2, 301), (7, 1000, 2, 64), (5, 100, 2, 500), (10, 1000, 1, 40), (10, 1000, 0, 100)])
def test_forward_packed_matches_oracle(cfg):
    """fused decode+forward from the 2-bit genome == oracle(encoders(oracle)) on the same sites, both strands,
    chromosome edges and N runs included."""
    from mural_amd.data import PackedGenome
    r, R, model_no, n_sites = cfg
    rng = np.random.default_rng(1000 + r + R + model_no)
    n = 60_000
    raw = rng.choice(np.frombuffer(b"ACGTN", np.uint8), size=n, p=[.248, .248, .248, .248, .008])
    raw[30000:30050] = ord("N")
    seq = raw.tobytes().decode()
    codes = encode_ref.seq_to_codes(seq)
    pos = rng.integers(0, n, size=n_sites)
    pos[:6] = [0, 1, 3, n - 1, n - 2, 30010]
    strand = rng.integers(0, 2, size=n_sites).astype(np.uint8)
    sym = ["-" if s else "+" for s in strand]
    orc = snv_ref.build(model_no, local_radius=r, distal_radius=R)
    sd = synth.synth_state_dict(orc.state_dict(), 77)
    orc.load_state_dict(sd)
    orc.eval()
    hp = np.array([r, 3, R, 150, 75, 32, 3, 4, model_no])
    model, _ = product_from_hp(hp)
    model.load_state_dict(sd)
    model = model.cuda().eval()
    cat = torch.from_numpy(encode_ref.kmer_encode(codes, pos, sym, r, 3))
    x = torch.from_numpy(encode_ref.onehot_encode(codes, pos, sym, R))
    with torch.no_grad():
        want = orc((torch.zeros(n_sites, 1, dtype=torch.float64), cat), x).numpy()
    genome = PackedGenome.from_sequence(seq, "cuda")
    got = model.forward_packed(genome, torch.from_numpy(pos).cuda(), torch.from_numpy(strand).cuda(), local_radius=r,
                               local_order=3).cpu().numpy()
    assert_probs_close(got, want, model_no, str(cfg))


@pytest.mark.parametrize("R", [100, 1000])
def test_forward_packed_iupac_windows_match_reference_encoding(R):
    """The G1 string holds R Y M S W K B D H V: the packed path must evaluate them as the reference's fractional one-hot columns
    (preprocessing.py:762-772), not as N.  Expected = oracle model on the reference's own encoder output (R = 100: the golden
    tensors of seq_ohe_encoder themselves; R = 1000: the oracle encoder, which the golden check sums pin)."""
    from mural_amd.data import PackedGenome
    fx = U.load("encode.npz")
    seq = fx["seq"].tobytes().decode()
    codes = encode_ref.seq_to_codes(seq)
    r = 7
    orc = snv_ref.build(2, local_radius=r, distal_radius=R)
    sd = synth.synth_state_dict(orc.state_dict(), 91)
    orc.load_state_dict(sd)
    orc.eval()
    model, _ = product_from_hp(np.array([r, 3, R, 150, 75, 32, 3, 4, 2]))
    model.load_state_dict(sd)
    model = model.cuda().eval()
    genome = PackedGenome.from_sequence(seq, "cuda")
    n_frac = 0
    for neg in (False, True):
        tag = "neg" if neg else "pos"
        starts = fx["starts"][fx["strands"].astype(bool) == neg]
        sym = ["-" if neg else "+"] * len(starts)
        x = fx[f"ohe_snv_{tag}_R{R}"] if R == 100 else encode_ref.onehot_encode(codes, starts, sym, R)
        cat = fx[f"kmer_snv_{tag}_r{r}_k3"]
        with torch.no_grad():
            want = orc((torch.zeros(len(starts), 1, dtype=torch.float64), torch.from_numpy(cat)), torch.from_numpy(x)).numpy()
            as_n = orc((torch.zeros(len(starts), 1, dtype=torch.float64), torch.from_numpy(cat)),
                       torch.from_numpy(encode_ref.onehot_encode(np.where(codes > 4, 4, codes), starts, sym, R))).numpy()
        got = model.forward_packed(genome, torch.from_numpy(starts).cuda(),
                                   torch.full((len(starts),), int(neg), dtype=torch.uint8).cuda(), local_radius=r,
                                   local_order=3).cpu().numpy()
        assert_probs_close(got, want, 2, f"iupac {tag} R{R}")
        n_frac += int((np.abs(np.exp(want) - np.exp(as_n)).max(axis=1) > 1e-4).sum())
    assert n_frac >= 3        # the fixture does distinguish "IUPAC as fractions" from "IUPAC as N"


def test_forward_packed_many_ambiguity_codes():
    """A genome with thousands of IUPAC codes: every window overlaps several side-table entries (multi-round wave search)."""
    from mural_amd.data import PackedGenome
    rng = np.random.default_rng(12)
    n, r, R, n_sites = 50_000, 10, 1000, 150
    raw = rng.choice(np.frombuffer(b"ACGTNRYMSWKBDHV", np.uint8), size=n, p=[.23, .23, .23, .23, .01] + [.007] * 10)
    seq = raw.tobytes().decode()
    codes = encode_ref.seq_to_codes(seq)
    pos = rng.integers(0, n, size=n_sites)
    pos[:4] = [0, 2, n - 1, n - 3]
    strand = rng.integers(0, 2, size=n_sites).astype(np.uint8)
    sym = ["-" if s else "+" for s in strand]
    orc = snv_ref.build(2, local_radius=r, distal_radius=R)
    sd = synth.synth_state_dict(orc.state_dict(), 92)
    orc.load_state_dict(sd)
    orc.eval()
    model, _ = product_from_hp(np.array([r, 3, R, 150, 75, 32, 3, 4, 2]))
    model.load_state_dict(sd)
    model = model.cuda().eval()
    with torch.no_grad():
        want = orc((torch.zeros(n_sites, 1, dtype=torch.float64), torch.from_numpy(encode_ref.kmer_encode(codes, pos, sym, r, 3))),
                   torch.from_numpy(encode_ref.onehot_encode(codes, pos, sym, R))).numpy()
    genome = PackedGenome.from_sequence(seq, "cuda")
    got = model.forward_packed(genome, torch.from_numpy(pos).cuda(), torch.from_numpy(strand).cuda(), local_radius=r,
                               local_order=3).cpu().numpy()
    assert_probs_close(got, want, 2, "dense ambiguity")


def test_rejects_non_encoding_input():
    fx = U.load("snv_synth_T_net2.npz")
    model, _ = product_from_hp(fx["hp"])
    model = model.cuda().eval()
    x = torch.rand(4, 4, 201, device="cuda")
    # the check does not drain the device: the call's output is NaN and the error surfaces at check_encoding() / a later call
    out = model((torch.zeros(4, 1).cuda(), torch.zeros(4, 9, dtype=torch.long).cuda()), x)
    assert torch.isnan(out).all()
    with pytest.raises(ValueError, match="not a MuRaL"):
        model.check_encoding()
    good = U.onehot(np.zeros((4, 201), np.uint8)).cuda()
    out = model((torch.zeros(4, 1).cuda(), torch.zeros(4, 9, dtype=torch.long).cuda()), good)
    model.check_encoding()
    assert torch.isfinite(out).all()
    model((torch.zeros(4, 1).cuda(), torch.zeros(4, 9, dtype=torch.long).cuda()), x)
    torch.cuda.synchronize()
    with pytest.raises(ValueError, match="not a MuRaL"):      # picked up without blocking by the next forward
        for _ in range(3):
            model((torch.zeros(4, 1).cuda(), torch.zeros(4, 9, dtype=torch.long).cuda()), good)
            torch.cuda.synchronize()
    with pytest.raises(AssertionError):
        model((torch.zeros(4, 1).cuda(), torch.zeros(4, 9, dtype=torch.long).cuda()), torch.zeros(4, 4, 199).cuda())


def test_cpu_tensors_fail_loudly():
    fx = U.load("snv_synth_T_net2.npz")
    model, _ = product_from_hp(fx["hp"])
    model = model.cuda().eval()
    with pytest.raises(RuntimeError):
        model((torch.zeros(4, 1), torch.zeros(4, 9, dtype=torch.long)), torch.zeros(4, 4, 201))


def test_reload_weights_invalidates_folded_copy():
    fx = U.load("snv_synth_T_net2.npz")
    model, _ = product_from_hp(fx["hp"])
    orc = U.snv_oracle_from_hp(fx["hp"])
    model = model.cuda().eval()
    orc.eval()
    cat = torch.from_numpy(fx["cat"])
    x = U.onehot(fx["codes"])
    for seed in (5, 6):
        sd = synth.synth_state_dict(orc.state_dict(), seed)
        orc.load_state_dict(sd)
        model.load_state_dict(sd)
        with torch.no_grad():
            want = orc((torch.zeros(len(cat), 1, dtype=torch.float64), cat), x).numpy()
            got = model((torch.zeros(len(cat), 1).cuda(), cat.cuda()), x.cuda()).cpu().numpy()
        assert_probs_close(got, want, 2, f"seed {seed}")


def test_model_predict_m_contract():
    from mural_amd.model import model_predict_m
    fx = U.load("predict_m.npz")
    model, _ = product_from_hp(fx["hp"])
    orc = U.snv_oracle_from_hp(fx["hp"])
    model.load_state_dict(U.snv_state_for(fx, orc))
    x = U.onehot(fx["codes"])
    batches, o = [], 0
    for n in fx["sizes"].tolist():
        batches.append((torch.from_numpy(fx["y"][o:o + n]), torch.zeros(n, 1, dtype=torch.float64),
                        torch.from_numpy(fx["cat"][o:o + n]), x[o:o + n]))
        o += n
    pred, total = model_predict_m(model, batches, nn.CrossEntropyLoss(reduction="sum"), torch.device("cuda"), 4, True, "snv")
    assert pred.is_cuda and pred.shape == fx["pred"].shape
    assert_probs_close(pred.cpu().numpy(), fx["pred"], 2, "predict_m")
    assert abs(total - float(fx["total_loss"])) <= 1e-4 * abs(float(fx["total_loss"]))
    # host batches took the host-classified symbol route (1 byte per column over PCIe); the same batches on the device take the dense
    # entry: same rows bit for bit, same loss; small fuse_rows: several flushes through the two staging buffers
    from mural_amd.model import nn_utils
    assert nn_utils._HostSymbolRoute.model_ok(model, "snv", True)
    taken = []
    begin = nn_utils._HostSymbolRoute.begin
    nn_utils._HostSymbolRoute.begin = lambda self, pending, rows: taken.append(rows) or begin(self, pending, rows)
    try:
        pred_h, _ = model_predict_m(model, batches, nn.CrossEntropyLoss(reduction="sum"), torch.device("cuda"), 4, True, "snv")
    finally:
        nn_utils._HostSymbolRoute.begin = begin
    assert taken == [len(fx["pred"])] and torch.equal(pred_h, pred)
    dev_batches = [tuple(t.cuda() for t in b) for b in batches]
    pred_d, total_d = model_predict_m(model, dev_batches, nn.CrossEntropyLoss(reduction="sum"), torch.device("cuda"), 4, True, "snv")
    assert torch.equal(pred, pred_d) and total == total_d
    pred_s, total_s = model_predict_m(model, batches, nn.CrossEntropyLoss(reduction="sum"), torch.device("cuda"), 4, True, "snv", fuse_rows=3)
    assert torch.equal(pred, pred_s) and abs(total - total_s) <= 1e-6 * abs(total)      # (float32 loss sums per flush)
    # a column that is no encoding is refused at once on this route
    bad = [tuple(t.clone() for t in b) for b in batches]
    bad[1][3][0, 2, 5] = 0.7
    with pytest.raises(ValueError, match="not a MuRaL"):
        model_predict_m(model, bad, nn.CrossEntropyLoss(reduction="sum"), torch.device("cuda"), 4, True, "snv")


def test_predict_bed_from_fasta_matches_oracle(tmp_path):
    """File-level path (C++ FASTA packer + BED reader + bed_reader row order + fused decode/forward) vs the oracle fed by
    the oracle's own encoders on the same sites; mixed focal bases are rejected like the reference does."""
    from mural_amd.data import ingest
    r, R = 5, 100
    rng = np.random.default_rng(4242)
    seqs = {}
    for name, n in (("chr2L", 4000), ("chrX", 2500)):
        raw = rng.choice(np.frombuffer(b"ACGTN", np.uint8), size=n, p=[.248, .248, .248, .248, .008])
        seqs[name] = raw.tobytes().decode()
    fa = tmp_path / "g.fa"
    fa.write_text("".join(f">{k} test\n" + "\n".join(s[i:i + 70] for i in range(0, len(s), 70)) + "\n" for k, s in seqs.items()))
    rows = []
    for name, s in seqs.items():                       # A sites on '+', T sites on '-': one focal base after complement
        arr = np.frombuffer(s.encode(), np.uint8)
        for p in np.sort(rng.choice(len(s), size=180, replace=False)):
            if arr[p] == ord("A"):
                rows.append((name, int(p), "+"))
            elif arr[p] == ord("T"):
                rows.append((name, int(p), "-"))
    bed = tmp_path / "s.bed"
    bed.write_text("".join(f"{c}\t{p}\t{p + 1}\t.\t{i % 4}\t{st}\n" for i, (c, p, st) in enumerate(rows)))
    orc = snv_ref.build(2, local_radius=r, distal_radius=R)
    sd = synth.synth_state_dict(orc.state_dict(), 78)
    orc.load_state_dict(sd)
    orc.eval()
    model, _ = product_from_hp(np.array([r, 3, R, 150, 75, 32, 3, 4, 2]))
    model.load_state_dict(sd)
    res = ingest.predict_bed(model, fa, bed, local_radius=r, local_order=3, segment_center=500)
    assert len(res["start"]) == len(rows)
    # reference row order: per 500-bp segment, '+' rows then '-' rows (bed_reader); recompute it independently
    from mural_amd.data.batching import segment_order
    chrom = np.array([c for c, _, _ in rows])
    start = np.array([p for _, p, _ in rows])
    neg = np.array([st == "-" for _, _, st in rows])
    order, _ = segment_order(chrom, start, neg, 500)
    assert np.array_equal(res["start"], start[order]) and np.array_equal(res["chrom"], chrom[order])
    assert np.array_equal(res["strand"] == "-", neg[order])
    assert np.array_equal(res["label"], (np.arange(len(rows)) % 4)[order].astype(np.float32))
    want = np.zeros((len(rows), 4))
    for name, s in seqs.items():
        sel = np.nonzero(chrom[order] == name)[0]
        codes = encode_ref.seq_to_codes(s)
        sym = ["-" if v else "+" for v in neg[order][sel]]
        cat = torch.from_numpy(encode_ref.kmer_encode(codes, start[order][sel], sym, r, 3))
        x = torch.from_numpy(encode_ref.onehot_encode(codes, start[order][sel], sym, R))
        with torch.no_grad():
            want[sel] = torch.softmax(orc((torch.zeros(len(sel), 1, dtype=torch.float64), cat), x), dim=1).numpy()
    assert np.abs(res["prob"] - want).max() <= PROB_TOL
    # a C site inside the first '+' group: the reference exits with "different bases", here ValueError
    arr = np.frombuffer(seqs["chr2L"].encode(), np.uint8)
    p0 = next(p for c, p, st in rows if c == "chr2L" and st == "+")
    cpos = int(p0 + 1 + np.nonzero(arr[p0 + 1:p0 + 400] == ord("C"))[0][0])
    rows2 = sorted(rows + [("chr2L", cpos, "+")], key=lambda t: (t[0] != "chr2L", t[1]))
    bed2 = tmp_path / "mixed.bed"
    bed2.write_text("".join(f"{c}\t{p}\t{p + 1}\t.\t0\t{st}\n" for c, p, st in rows2))
    with pytest.raises(ValueError, match="different bases"):
        ingest.predict_bed(model, fa, bed2, local_radius=r, local_order=3, segment_center=500)


def test_packed_path_other_channel_and_kernel_sizes():
    """Shapes outside the fused kernels (16 channels, k=5) take the per-layer path; packed-genome input equals dense input."""
    from mural_amd.data import PackedGenome
    from mural_amd.data.genome import pack_sequence
    fx = U.load("snv_synth_generic_c16k5_net2.npz")
    model, _ = product_from_hp(fx["hp"])
    orc = U.snv_oracle_from_hp(fx["hp"])
    sd = U.snv_state_for(fx, orc)
    model.load_state_dict(sd)
    orc.load_state_dict(sd)
    orc.eval()
    model = model.cuda().eval()
    assert model._fused_ok() is False
    rng = np.random.default_rng(77)
    seq = rng.choice(np.frombuffer(b"ACGTN", np.uint8), size=5000, p=[.245, .245, .245, .245, .02]).tobytes().decode()
    packed, mask, n, amb = pack_sequence(seq)
    genome = PackedGenome(packed, mask, n, "cuda", amb)
    pos = torch.from_numpy(rng.integers(0, n, size=40)).cuda()
    strand = torch.from_numpy(rng.integers(0, 2, size=40).astype(np.uint8)).cuda()
    r, R = int(fx["hp"][0]), int(fx["hp"][2])
    with torch.no_grad():
        got = model.forward_packed(genome, pos, strand, local_radius=r, local_order=3)
        x = genome.encode_onehot(pos, strand, R)
        cat = genome.encode_kmer(pos, strand, r, 3)
        want = model((torch.zeros(40, 1, dtype=torch.float64, device="cuda"), cat), x)
        ref = orc((torch.zeros(40, 1, dtype=torch.float64), cat.cpu()), x.cpu())
    assert torch.equal(got, want)
    assert_probs_close(got.cpu().numpy(), ref.numpy(), 2, "generic packed")
    # the per-layer path caches the folded BatchNorm affines and the re-laid-out conv weights per module: an in-place update of a
    # weight, of running statistics, or a load_state_dict must show in the next forward (the caches key on the tensors' versions)
    with torch.no_grad():
        model.conv1[1].weight.mul_(1.25)
        model.RBs1[0].bn1.running_var.add_(0.5)
        model.distal_fc1[0].bias.add_(0.1)
    sd2 = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    orc.load_state_dict(sd2)
    with torch.no_grad():
        got2 = model((torch.zeros(40, 1, dtype=torch.float64, device="cuda"), cat), x)
        ref2 = orc((torch.zeros(40, 1, dtype=torch.float64), cat.cpu()), x.cpu())
    assert float((got2 - want).abs().max()) > 1e-4
    assert_probs_close(got2.cpu().numpy(), ref2.numpy(), 2, "generic, after in-place updates")
    model.load_state_dict(sd)
    with torch.no_grad():
        got3 = model((torch.zeros(40, 1, dtype=torch.float64, device="cuda"), cat), x)
    assert torch.equal(got3, want)


def test_train_batches_from_files_match_reference_pipeline_order(tmp_path):
    """FASTA + BED -> training batches: rows, labels, k-mer columns and one-hot windows equal the oracle encoders applied to the
    sites in bed_reader order, cut by generate_data_batches (2 segments per group, batch 7, tail rows carried forward)."""
    from mural_amd.data import ingest
    from mural_amd.data.batching import segment_order
    r, R = 4, 60
    rng = np.random.default_rng(99)
    seqs = {"chrA": rng.choice(np.frombuffer(b"ACGTN", np.uint8), size=3000, p=[.24, .25, .25, .24, .02]).tobytes().decode(),
            "chrB": rng.choice(np.frombuffer(b"ACGT", np.uint8), size=1500).tobytes().decode()}
    fa = tmp_path / "g.fa"
    fa.write_text("".join(f">{k}\n" + "\n".join(s[i:i + 60] for i in range(0, len(s), 60)) + "\n" for k, s in seqs.items()))
    rows = [(name, int(p), "+-"[int(rng.integers(0, 2))], int(rng.integers(0, 4))) for name, s in seqs.items()
            for p in np.sort(rng.choice(len(s), size=45, replace=False))]
    bed = tmp_path / "s.bed"
    bed.write_text("".join(f"{c}\t{p}\t{p + 1}\t.\t{lab}\t{st}\n" for c, p, st, lab in rows))
    chrom = np.array([c for c, _, _, _ in rows])
    start = np.array([p for _, p, _, _ in rows])
    neg = np.array([st == "-" for _, _, st, _ in rows])
    lab = np.array([v for _, _, _, v in rows], dtype=np.float32)
    order, _ = segment_order(chrom, start, neg, 400)
    want_cat, want_x = [], []
    for i in order:
        codes = encode_ref.seq_to_codes(seqs[chrom[i]])
        sym = ["-" if neg[i] else "+"]
        want_cat.append(encode_ref.kmer_encode(codes, start[i:i + 1], sym, r, 3)[0])
        want_x.append(encode_ref.onehot_encode(codes, start[i:i + 1], sym, R)[0])
    want_cat, want_x, want_y = np.stack(want_cat), np.stack(want_x), lab[order]
    got = list(ingest.train_batches_from_files(fa, bed, 7, r, 3, R, segment_center=400, sampled_segments=2, shuffle=False))
    assert sum(b[0].shape[0] for b in got) == len(rows)
    assert all(b[0].shape[0] == 7 for b in got[:-1]) and got[0][1].dtype == torch.float64 and got[0][1].shape == (7, 1)
    # without shuffling the batches walk the bed_reader order (a carried tail goes first in its next group: still in order)
    y = torch.cat([b[0] for b in got]).cpu().numpy().reshape(-1)
    cat = torch.cat([b[2] for b in got]).cpu().numpy()
    x = torch.cat([b[3] for b in got]).cpu().numpy()
    assert np.array_equal(y, want_y) and np.array_equal(cat, want_cat) and np.array_equal(x, want_x)
    # shuffled: the same multiset of rows
    g = torch.Generator().manual_seed(1)
    sh = list(ingest.train_batches_from_files(fa, bed, 7, r, 3, R, segment_center=400, sampled_segments=2, shuffle=True, generator=g))
    cat2 = torch.cat([b[2] for b in sh]).cpu().numpy()
    assert sorted(map(tuple, cat2.tolist())) == sorted(map(tuple, want_cat.tolist())) and not np.array_equal(cat2, want_cat)


@pytest.mark.parametrize("name", ["snv_synth_S_net2.npz", "snv_synth_S_net0.npz", "snv_synth_R300_net2_c3.npz", "snv_pretrained_human_AT.npz"])
def test_local_branch_valu_kernel_on_golden_batches(name, monkeypatch):
    """The local branch runs on the MFMA kernel whenever its weights fit LDS; the VALU kernel (the fallback for wider layers)
    is forced here on the same golden inputs (ragged last tile, 3- and 4-class heads, the 13-column pretrained shape,
    Network0's raw logits)."""
    if name not in SNV_FORWARD:
        pytest.skip("fixture not present")
    monkeypatch.setenv("MURAL_DEBUG_LOCAL_VALU", "1")
    test_forward_dense_matches_reference(name)


def test_local_branch_kernels_agree_on_a_large_batch():
    """The MFMA kernel (default) must agree with the VALU kernel (MURAL_DEBUG_LOCAL_VALU) to fp32 rounding on a large batch."""
    fx = U.load("snv_synth_S_net0.npz")
    model, _ = product_from_hp(fx["hp"])
    orc = U.snv_oracle_from_hp(fx["hp"])
    model.load_state_dict(U.snv_state_for(fx, orc))
    model = model.cuda().eval()
    rng = np.random.default_rng(11)
    n = 4096 * 3 + 17
    cat = torch.from_numpy(rng.integers(0, 65, size=(n, int(fx["cat"].shape[1])))).cuda()
    x = torch.zeros((n, 4, 2001), device="cuda")
    cont = torch.zeros(n, 1, dtype=torch.float64, device="cuda")
    with torch.no_grad():
        a = model((cont, cat), x)
        os.environ["MURAL_DEBUG_LOCAL_VALU"] = "1"
        try:
            b = model((cont, cat), x)
        finally:
            del os.environ["MURAL_DEBUG_LOCAL_VALU"]
    assert (a - b).abs().max().item() <= 2e-5 * max(1.0, b.abs().max().item())


@pytest.mark.parametrize("h1,h2,r,nc", [(40, 24, 7, 4), (16, 16, 3, 2), (200, 100, 10, 6)])
def test_local_branch_other_widths_on_both_kernels(h1, h2, r, nc, monkeypatch):
    """Local branch with other hidden widths / column counts / class counts (block grids 3x2, 1x1, 13x7: the last one does not
    fit LDS and stays on the VALU kernel) against the oracle, on the default kernel and with the VALU kernel forced."""
    hp = np.array([r, 3, 300, h1, h2, 32, 3, nc, 0])
    model, _ = product_from_hp(hp)
    orc = U.snv_oracle_from_hp(hp)
    sd = synth.synth_state_dict(orc.state_dict(), 321)
    model.load_state_dict(sd)
    orc.load_state_dict(sd)
    model = model.cuda().eval()
    orc.eval()
    rng = np.random.default_rng(5)
    n = 77
    cat = rng.integers(0, 65, size=(n, 2 * r + 1 - 2)).astype(np.int64)
    x = torch.zeros((n, 4, 601))
    cont = torch.zeros(n, 1, dtype=torch.float64)
    with torch.no_grad():
        want = orc((cont, torch.from_numpy(cat)), x).numpy()
        got_mfma = model((cont.cuda(), torch.from_numpy(cat).cuda()), x.cuda()).cpu().numpy()
        monkeypatch.setenv("MURAL_DEBUG_LOCAL_VALU", "1")
        got_valu = model((cont.cuda(), torch.from_numpy(cat).cuda()), x.cuda()).cpu().numpy()
    assert_probs_close(got_valu, want, 0, "valu")
    assert_probs_close(got_mfma, want, 0, "mfma")


# ------------------------------------------------------------------------------------------------ small-batch launches
@pytest.mark.parametrize("name", SNV_FORWARD)
def test_throughput_kernels_on_golden_batches(name, monkeypatch):
    """Calls of up to 256 sites take the latency-shaped launches (one workgroup per (site, tower), dense window and local branch
    inside the first-stage launch); the golden batches are small, so the throughput-shaped launches are forced on them here."""
    monkeypatch.setenv("MURAL_DEBUG_NO_SMALL_BATCH", "1")
    test_forward_dense_matches_reference(name)


@pytest.mark.parametrize("n_sites", [1, 16, 255, 256])
def test_small_batch_launches_equal_throughput_launches(n_sites, monkeypatch):
    """Same sites through both launch shapes, dense and packed entry, both strands, IUPAC codes and a chromosome edge: the
    per-site arithmetic is the same, so the results agree to the last bits."""
    from mural_amd.data import PackedGenome
    fx = U.load("snv_synth_S_net2.npz")
    model, _ = product_from_hp(fx["hp"])
    orc = U.snv_oracle_from_hp(fx["hp"])
    model.load_state_dict(U.snv_state_for(fx, orc))
    model = model.cuda().eval()
    r, R = int(fx["hp"][0]), int(fx["hp"][2])
    rng = np.random.default_rng(n_sites)
    raw = rng.choice(np.frombuffer(b"ACGTNRY", np.uint8), size=20_000, p=[.245, .245, .245, .245, .01, .005, .005])
    seq = raw.tobytes().decode()
    codes = encode_ref.seq_to_codes(seq)
    pos = rng.integers(0, len(seq), size=n_sites)
    pos[0] = 3
    strand = rng.integers(0, 2, size=n_sites).astype(np.uint8)
    sym = ["-" if s else "+" for s in strand]
    cat = torch.from_numpy(encode_ref.kmer_encode(codes, pos, sym, r, 3)).cuda()
    x = torch.from_numpy(encode_ref.onehot_encode(codes, pos, sym, R)).cuda()
    cont = torch.zeros(n_sites, 1, dtype=torch.float64, device="cuda")
    genome = PackedGenome.from_sequence(seq, "cuda")

    def both():
        with torch.no_grad():
            d = model((cont, cat), x).cpu().numpy()
            p = model.forward_packed(genome, torch.from_numpy(pos).cuda(), torch.from_numpy(strand).cuda(), local_radius=r,
                                     local_order=3).cpu().numpy()
        return d, p

    d_small, p_small = both()
    monkeypatch.setenv("MURAL_DEBUG_NO_SMALL_BATCH", "1")
    d_big, p_big = both()
    assert np.isfinite(d_small).all() and np.isfinite(p_small).all()
    assert np.abs(d_small - d_big).max() <= 2e-6 and np.abs(p_small - p_big).max() <= 2e-6
    assert np.abs(d_small - p_small).max() <= 2e-6


class _PoisonedTorch:
    """torch with empty() / empty_like() that hand out memory pre-filled with 0xFF bytes (NaN as float32, -1 as integers)."""

    def __getattr__(self, k):
        return getattr(torch, k)

    @staticmethod
    def empty(*a, **k):
        t = torch.empty(*a, **k)
        t.view(torch.uint8).fill_(255) if t.is_contiguous() and t.numel() else None
        return t

    @staticmethod
    def empty_like(*a, **k):
        t = torch.empty_like(*a, **k)
        t.view(torch.uint8).fill_(255) if t.is_contiguous() and t.numel() else None
        return t


@pytest.mark.parametrize("n_sites", [16, 700])
def test_forward_does_not_read_unwritten_workspace(n_sites, monkeypatch):
    """The workspace of a forward (k-mer ids, symbols, pooled first-layer rows, tower hand-over buffers, arrival counters, logits)
    comes from torch.empty: every part must be written before it is read.  The same call with the workspace and the output
    pre-filled with 0xFF bytes gives the same result, for both launch shapes and both entries."""
    from mural_amd.data import PackedGenome
    from mural_amd.model import model_snv as MS
    fx = U.load("snv_synth_S_net2.npz")
    model, _ = product_from_hp(fx["hp"])
    orc = U.snv_oracle_from_hp(fx["hp"])
    model.load_state_dict(U.snv_state_for(fx, orc))
    model = model.cuda().eval()
    r, R = int(fx["hp"][0]), int(fx["hp"][2])
    rng = np.random.default_rng(7)
    seq = rng.choice(np.frombuffer(b"ACGTN", np.uint8), size=20_000, p=[.247, .247, .247, .247, .012]).tobytes().decode()
    codes = encode_ref.seq_to_codes(seq)
    pos = rng.integers(0, len(seq), size=n_sites)
    strand = rng.integers(0, 2, size=n_sites).astype(np.uint8)
    sym = ["-" if s else "+" for s in strand]
    cat = torch.from_numpy(encode_ref.kmer_encode(codes, pos, sym, r, 3)).cuda()
    x = torch.from_numpy(encode_ref.onehot_encode(codes, pos, sym, R)).cuda()
    cont = torch.zeros(n_sites, 1, dtype=torch.float64, device="cuda")
    genome = PackedGenome.from_sequence(seq, "cuda")

    def both():
        with torch.no_grad():
            d = model((cont, cat), x).cpu().numpy()
            p = model.forward_packed(genome, torch.from_numpy(pos).cuda(), torch.from_numpy(strand).cuda(), local_radius=r,
                                     local_order=3).cpu().numpy()
        return d, p

    d0, p0 = both()
    monkeypatch.setattr(MS, "torch", _PoisonedTorch())
    model._ws = None
    model._ws_rows = [0, 0]
    d1, p1 = both()
    assert np.isfinite(d1).all() and np.isfinite(p1).all()
    assert np.abs(d1 - d0).max() <= 2e-6 and np.abs(p1 - p0).max() <= 2e-6


@pytest.mark.parametrize("n_sites", [16, 200, 700])
def test_forward_writes_stay_inside_their_workspace_regions(n_sites, monkeypatch):
    """MURAL_DEBUG_WS_GUARD puts 4 KB of unused bytes behind every region of the forward's workspace (k-mer ids, symbols, pooled
    rows, logits, hand-over tiles / arrival counters; ``mural_debug_last_ws_layout`` lists them): with the workspace poisoned before the
    call, every guard byte is still the
    poison afterwards -- no kernel of either launch shape writes outside the region it was given (dense and packed entry)."""
    import ctypes as C
    from mural_amd import _lib
    from mural_amd.data import PackedGenome
    from mural_amd.model import model_snv as MS
    guard = 4096
    monkeypatch.setenv("MURAL_DEBUG_WS_GUARD", str(guard))
    monkeypatch.setattr(MS, "torch", _PoisonedTorch())
    fx = U.load("snv_synth_S_net2.npz")
    model, _ = product_from_hp(fx["hp"])
    orc = U.snv_oracle_from_hp(fx["hp"])
    model.load_state_dict(U.snv_state_for(fx, orc))
    model = model.cuda().eval()
    r, R = int(fx["hp"][0]), int(fx["hp"][2])
    rng = np.random.default_rng(n_sites)
    seq = rng.choice(np.frombuffer(b"ACGTN", np.uint8), size=20_000, p=[.247, .247, .247, .247, .012]).tobytes().decode()
    codes = encode_ref.seq_to_codes(seq)
    pos = rng.integers(0, len(seq), size=n_sites)
    strand = rng.integers(0, 2, size=n_sites).astype(np.uint8)
    sym = ["-" if s else "+" for s in strand]
    cat = torch.from_numpy(encode_ref.kmer_encode(codes, pos, sym, r, 3)).cuda()
    x = torch.from_numpy(encode_ref.onehot_encode(codes, pos, sym, R)).cuda()
    cont = torch.zeros(n_sites, 1, dtype=torch.float64, device="cuda")
    genome = PackedGenome.from_sequence(seq, "cuda")
    for dense in (1, 0):
        model._ws = None
        model._ws_rows = [0, 0]
        with torch.no_grad():
            if dense:
                out = model((cont, cat), x)
            else:
                out = model.forward_packed(genome, torch.from_numpy(pos).cuda(), torch.from_numpy(strand).cuda(), local_radius=r, local_order=3)
        assert torch.isfinite(out).all()
        layout = (C.c_size_t * 64)()
        n_regions = _lib.lib().mural_debug_last_ws_layout(layout, 32)      # the carve of the call just made (this thread's latest)
        assert n_regions == 8
        ws = model._ws.cpu().numpy()
        for i in range(n_regions):
            off, size = layout[2 * i], layout[2 * i + 1]
            zone = ws[off + size:off + size + guard]
            assert len(zone) == guard and (zone == 255).all(), f"region {i} (dense={dense}): a kernel wrote behind its {size} bytes"


def test_long_window_writes_stay_inside_their_workspace_regions(monkeypatch):
    """The same guard check for the segmented long-window path (R = 2000: ten regions -- the eight of the regular forward plus the
    segments' pooled outputs of both kinds; the segments themselves are read in place from x0 since round 6): no segment / scatter
    launch writes outside its region, and the poisoned workspace does not leak into the result (the per-layer path on the same inputs
    agrees)."""
    import ctypes as C
    from mural_amd import _lib
    from mural_amd.data import PackedGenome
    from mural_amd.model import model_snv as MS
    from mural_amd.model import generic_eval
    guard = 4096
    monkeypatch.setenv("MURAL_DEBUG_WS_GUARD", str(guard))
    monkeypatch.setattr(MS, "torch", _PoisonedTorch())
    r, R, n_sites = 7, 2000, 300
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
    genome = PackedGenome.from_sequence(seq, "cuda")
    assert model._fused_ok()
    with torch.no_grad():
        want = generic_eval.forward(model, cat, x, MS.POOLS_MID, MS.POOLS_LARGE).cpu().numpy()
    for dense in (1, 0):
        model._ws = None
        model._ws_rows = [0, 0]
        with torch.no_grad():
            if dense:
                out = model((cont, cat), x)
            else:
                out = model.forward_packed(genome, torch.from_numpy(pos).cuda(), torch.from_numpy(strand).cuda(), local_radius=r, local_order=3)
        assert np.abs(out.cpu().numpy() - want).max() <= 1e-4      # (log-probabilities; fp32 summation orders differ between the paths)
        layout = (C.c_size_t * 64)()
        n_regions = _lib.lib().mural_debug_last_ws_layout(layout, 32)
        assert n_regions == 10
        ws = model._ws.cpu().numpy()
        for i in range(n_regions):
            off, size = layout[2 * i], layout[2 * i + 1]
            zone = ws[off + size:off + size + guard]
            assert len(zone) == guard and (zone == 255).all(), f"region {i} (dense={dense}): a kernel wrote behind its {size} bytes"


def test_front_only_long_windows_in_chunks():
    """R = 16000 (front-only handle with a fused mid tower): more sites than one front / finish pair takes (4096) go through in chunks on
    one workspace -- every site's result is the one it gets in a call of its own block (per-site results do not depend on the batch)."""
    from mural_amd.data import PackedGenome
    R, r = 16000, 7
    rng = np.random.default_rng(3)
    seq = rng.choice(np.frombuffer(b"ACGTN", np.uint8), size=60_000, p=[.248, .248, .248, .248, .008]).tobytes().decode()
    genome = PackedGenome.from_sequence(seq, "cuda")
    model, _ = product_from_hp(np.array([r, 3, R, 150, 75, 32, 3, 4, 2]))
    model.load_state_dict(synth.synth_state_dict(model.state_dict(), 11))
    model = model.cuda().eval()
    n = 4096 + 300
    pos = torch.from_numpy(rng.integers(0, 60_000, size=n)).cuda()
    strand = torch.from_numpy(rng.integers(0, 2, size=n).astype(np.uint8)).cuda()
    with torch.no_grad():
        whole = model.forward_packed(genome, pos, strand, local_radius=r, local_order=3)
        assert model._front_ok() and model._front_mid and model._front_chunk == 4096
        a = model.forward_packed(genome, pos[:1000], strand[:1000], local_radius=r, local_order=3)
        b = model.forward_packed(genome, pos[4000:], strand[4000:], local_radius=r, local_order=3)
    assert torch.isfinite(whole).all()
    assert float((whole[:1000] - a).abs().max()) <= 1e-5 and float((whole[4000:] - b).abs().max()) <= 1e-5


@pytest.mark.parametrize("R,model_no,fused", [(2000, 2, True), (4000, 2, True), (4000, 1, True), (3000, 2, True), (2000, 2, False), (4000, 1, False),
                                              (8000, 2, True), (16000, 2, False), (16000, 1, False)])
# (R = 16000: a FRONT-ONLY handle -- the packed entry runs stage 1 and the large tower's segmented first stage fused and finishes per layer,
# the dense entry stays per-layer)
def test_long_windows_match_oracle(R, model_no, fused, monkeypatch):
    """Windows beyond the shipped radius (the reference advertises inputs of up to 64 kb, CHANGELOG:13).  The pooled first-stage row of
    the large tower (267 / 400 / 534 columns at R = 2000 / 3000 / 4000) does not fit a wave's LDS image: the fused path runs that
    stage on segments with halo columns (MuralSnvModel::longwin, csrc/snv_model.hip) and the rest as usual; with
    MURAL_DEBUG_NO_LONGWIN the C side refuses the shape and eval takes the per-layer HIP path (model/generic_eval.py).  Packed and
    dense entry of both against the oracle fed by the oracle encoders, both strands, chromosome ends, a batch larger than one
    segment launch's wave count."""
    from mural_amd.data import PackedGenome
    if not fused and R < 15000:
        monkeypatch.setenv("MURAL_DEBUG_NO_LONGWIN", "1")
    r = 7
    rng = np.random.default_rng(R + model_no)
    n = 40_000 if R <= 4000 else 90_000
    raw = rng.choice(np.frombuffer(b"ACGTN", np.uint8), size=n, p=[.248, .248, .248, .248, .008])
    seq = raw.tobytes().decode()
    codes = encode_ref.seq_to_codes(seq)
    pos = rng.integers(0, n, size=24)
    pos[:4] = [0, 5, n - 1, n // 2]
    strand = rng.integers(0, 2, size=len(pos)).astype(np.uint8)
    sym = ["-" if s else "+" for s in strand]
    orc = snv_ref.build(model_no, local_radius=r, distal_radius=R)
    sd = synth.synth_state_dict(orc.state_dict(), 5 + R)
    orc.load_state_dict(sd)
    orc.eval()
    model, _ = product_from_hp(np.array([r, 3, R, 150, 75, 32, 3, 4, model_no]))
    model.load_state_dict(sd)
    model = model.cuda().eval()
    cat = torch.from_numpy(encode_ref.kmer_encode(codes, pos, sym, r, 3))
    x = torch.from_numpy(encode_ref.onehot_encode(codes, pos, sym, R))
    with torch.no_grad():
        want = orc((torch.zeros(len(pos), 1, dtype=torch.float64), cat), x).numpy()
        genome = PackedGenome.from_sequence(seq, "cuda")
        got = model.forward_packed(genome, torch.from_numpy(pos).cuda(), torch.from_numpy(strand).cuda(), local_radius=r,
                                   local_order=3).cpu().numpy()
        dense = model((torch.zeros(len(pos), 1, dtype=torch.float64).cuda(), cat.cuda()), x.cuda()).cpu().numpy()
    assert model._fused_ok() == fused and model._front_ok() == (R >= 15000)
    assert_probs_close(got, want, model_no, f"packed R={R}")
    assert_probs_close(dense, want, model_no, f"dense R={R}")
