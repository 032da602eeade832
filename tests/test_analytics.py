"""Validation-epoch analytics (mural_amd/evaluation.py + csrc/analytics.hip) against tests/golden/analytics.npz: numbers produced
by the reference's own freq_kmer_comp_multi / corr_calc_sub / Evaluator.evaluate_regional_score / ECELoss / ClasswiseECELoss /
BrierScore and by its Newton driver for the full-Dirichlet fit (oracle/make_golden.py g13).

CPU tests pin the oracle restatement (oracle/eval_ref.py) and the host half of the fit; the gpu tests run the HIP reductions.

Tolerances: correlations of group means 1e-9 (float64 sums in a different order); corr_calc_sub 2e-5 because the REFERENCE
accumulates float32 probabilities row by row in float32; NLL/ECE/Brier 2e-6 (the reference evaluates them in float32); fitted
weights 1e-5 relative and calibrated probabilities 1e-6 (float32 log features differ in the last bit between libms)."""
import numpy as np
import pytest
import torch

from mural_amd import evaluation as E
from oracle import eval_ref
from tests import _util as U

CASES = [("snv", "snv", 4, (3, 5, 7)), ("indel", "indel", 3, (2, 4, 6))]


def _fx(tag):
    fx = U.load("analytics.npz")
    return fx, fx[f"{tag}_codes"], fx[f"{tag}_label"], fx[f"{tag}_prob"], fx[f"{tag}_chrom"], fx[f"{tag}_start"]


def _sorted(chrom, start, *arrs):
    order = np.lexsort((start, chrom))             # the reference sorts by (chrom, start), stable w.r.t. ties -> same multiset
    return (chrom[order], start[order]) + tuple(a[order] for a in arrs)


# ---- oracle vs. the reference's numbers -------------------------------------------------------------------------------
@pytest.mark.parametrize("tag,model_type,nc,kmers", CASES)
def test_oracle_kmer_and_score(tag, model_type, nc, kmers):
    fx, codes, label, prob, chrom, start = _fx(tag)
    for k in kmers:
        got = eval_ref.freq_kmer_comp_multi(codes, label, prob, k, nc, model_type)
        assert np.allclose(got, fx[f"{tag}_kmer{k}"], atol=1e-12, equal_nan=True)
    corr, score, n_regions = eval_ref.regional_score(codes, label, prob, len(label), list(kmers[:2]), nc, model_type)
    assert n_regions == int(fx[f"{tag}_score"][1])
    assert abs(score - fx[f"{tag}_score"][0]) < 1e-9
    assert np.allclose(corr, fx[f"{tag}_score_corr"], atol=1e-9)


@pytest.mark.parametrize("tag,model_type,nc,kmers", CASES)
def test_oracle_regional_corr(tag, model_type, nc, kmers):
    fx, codes, label, prob, chrom, start = _fx(tag)
    c, s, lab, pr = _sorted(chrom, start, label, prob)
    for win in (1000, 5000):
        got = eval_ref.corr_calc_sub(c, s, lab, pr, win)
        assert np.allclose(got, fx[f"{tag}_win{win}"], atol=2e-5)


@pytest.mark.parametrize("tag,model_type,nc,kmers", CASES)
@pytest.mark.parametrize("dt", ["float32", "float64"])
def test_oracle_metrics(tag, model_type, nc, kmers, dt):
    fx, codes, label, prob, chrom, start = _fx(tag)
    got = eval_ref.calibration_metrics(fx[f"{tag}_metrics_prob_{dt}"], label)
    want = fx[f"{tag}_metrics_{dt}"]
    assert np.allclose([got["nll"], got["ece"], got["c_ece"], got["brier"]], want, atol=1e-7)


@pytest.mark.parametrize("tag,model_type,nc,kmers", CASES)
def test_oracle_and_host_fit(tag, model_type, nc, kmers):
    fx, codes, label, prob, chrom, start = _fx(tag)
    w, loss = eval_ref.fit_full_dirichlet(prob, label)
    assert np.abs(w - fx[f"{tag}_fit_w"]).max() < 1e-8 and abs(loss - float(fx[f"{tag}_fit_loss"])) < 1e-12
    # the product's host loop (Newton in the redundant parametrisation) fed with the oracle's row terms
    X_ = eval_ref.fit_features(prob)
    w2, loss2 = E.newton_full_dirichlet(lambda wt, need: eval_ref.fit_row_terms(X_, label, wt, need), nc)
    assert np.abs(w2 - fx[f"{tag}_fit_w"]).max() < 1e-8 and abs(loss2 - float(fx[f"{tag}_fit_loss"])) < 1e-12
    assert np.all(w2[-1] == 0)


@pytest.mark.parametrize("name", sorted(eval_ref.CALIBRATORS))
@pytest.mark.parametrize("tag,model_type,nc,kmers", CASES)
def test_oracle_and_host_fit_of_every_calibrator(tag, model_type, nc, kmers, name):
    """The calibrators calibrate_prob can be asked for (evaluation.py:303-316: FullDiri, FullDiriODIR, FullDiri1, FullDiri2, VectS,
    TempS) against the reference's own Newton driver and objective run under each parametrisation (G13; derivatives through the
    reference's linear _get_weights, finite-difference checked in oracle/make_golden.py)."""
    fx, codes, label, prob, chrom, start = _fx(tag)
    want_w, want_loss = fx[f"{tag}_fit_{name}_w"], float(fx[f"{tag}_fit_{name}_loss"])
    w, loss = eval_ref.fit_calibrator(prob, label, name)
    assert np.abs(w - want_w).max() < 1e-8 and abs(loss - want_loss) < 1e-12
    assert E.CALIBRATORS[name] == eval_ref.CALIBRATORS[name]
    method, ref_row, lam, mu, reg_norm = E.CALIBRATORS[name]
    if reg_norm:
        lam, mu = (lam / (nc * (nc + 1)), mu) if mu is None else (lam / (nc * (nc - 1)), mu / nc)
    X_ = eval_ref.fit_features(prob)
    w2, loss2 = E.newton_calibrator(lambda wt, need: eval_ref.fit_row_terms(X_, label, wt, need), nc, method, ref_row, lam, mu)
    assert np.abs(w2 - want_w).max() < 1e-8 and abs(loss2 - want_loss) < 1e-12
    if name == "TempS":                        # one temperature: a scaled identity minus its last row, no intercepts
        t = w2[0, 0]
        assert np.allclose(w2[:-1, :-1], t * (np.eye(nc)[:-1] - np.eye(nc)[-1]), atol=1e-12) and np.all(w2[:, -1] == 0)
    if name == "FullDiri2":
        assert np.any(w2[-1] != 0)             # no reference row: the last row is fitted too


def test_product_refuses_cpu_tensors():
    with pytest.raises(RuntimeError, match="HIP device"):
        E.freq_kmer_comp_multi(torch.zeros((4, 11), dtype=torch.int64), torch.zeros(4), torch.full((4, 4), 0.25), 3, 4)


def test_flank_columns():
    assert E._flank_columns(11, 3, "snv") == (4, 6, 1)
    assert E._flank_columns(11, 7, "snv") == (2, 6, 3)
    assert E._flank_columns(8, 4, "indel") == (2, 4, 2)
    assert eval_ref.flank_columns(11, 5, "snv") == [3, 4, 6, 7] and eval_ref.flank_columns(8, 2, "indel") == [3, 4]
    with pytest.raises(KeyError):
        E._flank_columns(5, 7, "snv")
    with pytest.raises(ValueError):
        E._flank_columns(10, 3, "snv")


# ---- the HIP path -----------------------------------------------------------------------------------------------------
def _dev(*arrs):
    return tuple(torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in arrs)


@pytest.mark.gpu
@pytest.mark.parametrize("tag,model_type,nc,kmers", CASES)
def test_gpu_kmer_and_score(tag, model_type, nc, kmers):
    fx, codes, label, prob, chrom, start = _fx(tag)
    d_codes, d_label, d_prob = _dev(codes, label, prob)
    for k in kmers:
        got = E.freq_kmer_comp_multi(d_codes, d_label, d_prob, k, nc, model_type)
        assert np.allclose(got, fx[f"{tag}_kmer{k}"], atol=1e-9, equal_nan=True), (k, got)
    corr, score, n_regions = E.regional_score(d_codes, d_label, d_prob, len(label), list(kmers[:2]), nc, model_type)
    assert n_regions == int(fx[f"{tag}_score"][1])
    assert abs(score - fx[f"{tag}_score"][0]) < 1e-6 * fx[f"{tag}_score"][0]     # float32-rounded group means inside (1 - r)^2
    assert np.allclose(corr, fx[f"{tag}_score_corr"], atol=2e-5)       # the reference's per-region mean is a float32 mean


@pytest.mark.gpu
@pytest.mark.parametrize("tag,model_type,nc,kmers", CASES)
def test_gpu_regional_corr(tag, model_type, nc, kmers):
    fx, codes, label, prob, chrom, start = _fx(tag)
    names = sorted(set(chrom.tolist()))
    cid = np.array([names.index(c) for c in chrom], dtype=np.int32)
    # row order must not matter (the kernel keys rows by window; the reference needs them sorted)
    perm = np.random.default_rng(0).permutation(len(cid))
    for order in (np.arange(len(cid)), perm, np.lexsort((start, cid))):
        d_cid, d_start, d_label, d_prob = _dev(cid[order], start[order], label[order], prob[order])
        for win in (1000, 5000):
            got = E.corr_calc_sub(d_cid, d_start, d_label, d_prob, win)
            assert np.allclose(got, fx[f"{tag}_win{win}"], atol=2e-5), (win, got)
    # float64 sums against an exact float64 evaluation of the same definition
    pr64 = prob.astype(np.float64)
    c, s, lab, pr = _sorted(chrom, start, label, pr64)
    want = eval_ref.corr_calc_sub(c, s, lab, pr, 1000)
    got = E.corr_calc_sub(*_dev(cid, start, label, pr64), 1000)
    assert np.allclose(got, want, atol=1e-10)
    # fewer than three windows -> 0, like the reference
    assert E.corr_calc_sub(*_dev(cid[:50] * 0, start[:50] % 100, label[:50], prob[:50]), 1000) == [0] * nc


@pytest.mark.gpu
@pytest.mark.parametrize("tag,model_type,nc,kmers", CASES)
@pytest.mark.parametrize("dt", ["float32", "float64"])
def test_gpu_metrics(tag, model_type, nc, kmers, dt):
    fx, codes, label, prob, chrom, start = _fx(tag)
    got = E.calibration_metrics(*_dev(fx[f"{tag}_metrics_prob_{dt}"], label))
    want = fx[f"{tag}_metrics_{dt}"]
    assert np.allclose([got["nll"], got["ece"], got["c_ece"], got["brier"]], want, atol=2e-6), (got, want)


@pytest.mark.gpu
@pytest.mark.parametrize("tag,model_type,nc,kmers", CASES)
def test_gpu_fit(tag, model_type, nc, kmers):
    fx, codes, label, prob, chrom, start = _fx(tag)
    d_prob, d_label = _dev(prob, label)
    # the device row terms at a non-trivial point against the oracle's float64 numpy evaluation
    rng = np.random.default_rng(3)
    w = np.hstack([np.eye(nc), np.zeros((nc, 1))]) + 0.1 * rng.standard_normal((nc, nc + 1))
    loss, g, h = E._fit_terms(d_prob, d_label.to(torch.int32), w, True)
    l0, g0, h0 = eval_ref.fit_row_terms(eval_ref.fit_features(prob), label, w, True)
    assert abs(loss - l0) < 1e-6 and np.abs(g - g0).max() < 1e-6 and np.abs(h - h0).max() < 1e-5
    wts, final = E.fit_full_dirichlet(d_prob, d_label)
    # the features are float32 logs (fulldirichlet.py:49-50): the device logf and numpy's differ in the last bit, and the
    # optimum of this nearly collinear problem (|w| up to ~65) moves by ~1e-6 relative; the calibrated map is what must agree
    want = fx[f"{tag}_fit_w"]
    assert np.abs(wts - want).max() < 1e-5 * np.abs(want).max(), np.abs(wts - want).max()
    assert abs(final - float(fx[f"{tag}_fit_loss"])) < 1e-9
    from mural_amd.calibration import dirichlet_calibrate
    assert np.abs(dirichlet_calibrate(prob, wts) - dirichlet_calibrate(prob, want)).max() < 1e-6
    weights, nll, prob_cal = E.calibrate_prob(d_prob, d_label, printer=lambda *a: None)
    assert abs(nll - final) < 1e-6 and prob_cal.shape == d_prob.shape
    # the other calibrators through the same device kernel: the calibrated probabilities are what must agree
    for name in ("FullDiriODIR", "FullDiri2", "VectS", "TempS"):
        wn, fn = E.fit_calibrator(d_prob, d_label, name)
        want_n = fx[f"{tag}_fit_{name}_w"]
        assert abs(fn - float(fx[f"{tag}_fit_{name}_loss"])) < 1e-8, name
        assert np.abs(dirichlet_calibrate(prob, wn) - dirichlet_calibrate(prob, want_n)).max() < 1e-6, name
    lines = []
    E.calibrate_prob(d_prob, d_label, printer=lambda *a: lines.append(a[0]), calibr_name="VectS")
    assert lines[0].startswith("Before VectS scaling") and lines[1].startswith("After VectS scaling")
    with pytest.raises(ValueError, match="unknown calibrator"):
        E.fit_calibrator(d_prob, d_label, "Platt")


@pytest.mark.gpu
def test_gpu_evaluator_and_errors():
    fx, codes, label, prob, chrom, start = _fx("snv")
    d_codes, d_label, d_prob = _dev(codes, label, prob)
    lines = []
    ev = E.Evaluator(d_codes, d_label, d_prob, 4, printer=lambda *a: lines.append(a))
    ev.evaluate_kmer([3, 5, 7])
    ev.evaluate_regional_score(len(label), [3, 5])
    assert lines[0][0] == "3mer correlation - all: " and abs(ev.metrics["score"] - fx["snv_score"][0]) < 1e-4
    bad = d_codes.clone()
    bad[5, 4] = 7
    with pytest.raises(ValueError, match="outside the expected range"):
        E.freq_kmer_comp_multi(bad, d_label, d_prob, 3, 4)
    with pytest.raises(ValueError, match="class label"):
        E.freq_kmer_comp_multi(d_codes, d_label + 3, d_prob, 3, 4)
    with pytest.raises(ValueError, match="every class"):
        E.fit_full_dirichlet(d_prob, torch.zeros_like(d_label))


@pytest.mark.gpu
def test_gpu_large_random_groups_match_numpy():
    """1M rows at k=7 (15625 groups, global-atomic path) and sorted windows (wave-uniform path): sums equal numpy's."""
    rng = np.random.default_rng(8)
    n, nc = 1_000_000, 4
    codes = rng.integers(0, 5, size=(n, 21)).astype(np.int64)
    label = rng.integers(0, nc, size=n)
    prob = rng.dirichlet([20, 1, 1, 1], size=n).astype(np.float32)
    d_codes, d_label, d_prob = _dev(codes, label, prob)
    keys, groups, status = E._kmer_keys(d_codes, 7, "snv")
    table = E._group_table(keys, d_label.to(torch.int32), d_prob, groups, status)
    cols = eval_ref.flank_columns(21, 7, "snv")
    k_np = np.zeros(n, np.int64)
    for c in cols:
        k_np = k_np * 5 + codes[:, c]
    assert np.array_equal(keys.cpu().numpy(), k_np)
    assert np.array_equal(table[:, 0], np.bincount(k_np, minlength=groups))
    for c in range(nc):
        assert np.array_equal(table[:, 1 + c], np.bincount(k_np, weights=(label == c), minlength=groups))
        assert np.allclose(table[:, 1 + nc + c], np.bincount(k_np, weights=prob[:, c].astype(np.float64), minlength=groups), rtol=1e-12)
    start = np.sort(rng.integers(0, 50_000_000, size=n))
    cid = np.zeros(n, np.int32)
    got = E.corr_calc_sub(*_dev(cid, start, label, prob), 100000)
    w = start // 100000
    cnt = np.bincount(w)
    live = cnt > 0
    want = []
    for c in range(nc):
        o = np.bincount(w, weights=(label == c))[live] / cnt[live]
        p = np.bincount(w, weights=prob[:, c].astype(np.float64))[live] / cnt[live]
        want.append(np.corrcoef(o, p)[0, 1])
    assert np.allclose(got, want, atol=1e-10)
