"""Validation-epoch analytics on the device.

Counterparts in the reference (MuRaL/evaluation/evaluation.py, driven from MuRaL/training.py:466-520 once per epoch):

  * ``freq_kmer_comp_multi`` (:48-67)        observed vs. predicted rate per flanking k-mer, Pearson r per class
  * ``corr_calc_sub`` (:124-193)              observed vs. predicted rate per genomic window, Pearson r per class
  * ``Evaluator`` (:489-587)                  evaluate_kmer / evaluate_regional_corr / evaluate_regional_score
  * ``calibrate_prob`` (:297-365)             full-Dirichlet fit + NLL / ECE / classwise ECE / Brier before and after

The reference walks pandas frames (``corr_calc_sub`` indexes ``data.loc[i, ...]`` once per row and per class) and fits the
calibrator with jax autodiff over an (n, m, m) outer-product tensor.  Here every pass over the rows is one HIP kernel of
``csrc/analytics.hip`` that reduces them into a small float64 table (group sums / histogram / loss-gradient-Hessian sums);
only the algebra on those tables (a Pearson r over at most a few thousand groups, a 20..72-parameter Newton step) runs on
the host.  Inputs are device tensors: ``local_codes`` is the order-1 local encoding the reference keeps in ``data_local``
(columns us_r..us_1, [mid,] ds_1..ds_r, values 0..4 -- ``PackedGenome.encode_kmer(pos, strand, r, 1)``), ``mut_type`` the
class labels, ``prob`` the (n, n_class) probabilities (float32 from the model, float64 after calibration).

There is no CPU fallback: tensors must live on a HIP device.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib

_FIT_MAX_CLASSES = 8


# ------------------------------------------------------------------------------------------------------------------
# device reductions
# ------------------------------------------------------------------------------------------------------------------
def _dev(t, what):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(f"{what} must be a tensor on a HIP device (mural_amd has no CPU path)")
    return t


def _prob(prob):
    prob = _dev(prob, "prob")
    if prob.dim() != 2:
        raise ValueError("prob must be (n, n_class)")
    if prob.dtype not in (torch.float32, torch.float64):
        prob = prob.to(torch.float32)
    return prob.contiguous()


def _label(mut_type, n):
    lab = _dev(mut_type, "mut_type").reshape(-1)
    if lab.shape[0] != n:
        raise ValueError(f"mut_type has {lab.shape[0]} rows, prob has {n}")
    return lab.to(torch.int32).contiguous()


def _raise_status(status, what):
    s = int(status.item())
    if s & 1:
        raise ValueError(f"{what}: code / chromosome / start outside the expected range")
    if s & 2:
        raise ValueError(f"{what}: class label outside [0, n_class)")


def _flank_columns(ncols, k, model_type):
    """(left0, right0, d) of the us_d..us_1 / ds_1..ds_d columns inside the order-1 local header
    (MuRaL/data/preprocessing.py:358-375: snv has a 'mid' column between the flanks, indel has none)."""
    if model_type not in ("snv", "indel"):
        raise ValueError(f"model_type {model_type} not supported!")
    d = k // 2
    r = (ncols - 1) // 2 if model_type == "snv" else ncols // 2
    if ncols != 2 * r + (1 if model_type == "snv" else 0):
        raise ValueError(f"local_codes has {ncols} columns: not an order-1 {model_type} local encoding")
    if d < 1 or d > r:
        raise KeyError(f"us{d}")          # the reference's column lookup fails the same way
    return r - d, (r + 1 if model_type == "snv" else r), d


def _kmer_keys(codes, k, model_type, region_size=0, n_regions=0):
    codes = _dev(codes, "local_codes")
    if codes.dim() != 2 or codes.dtype != torch.int64:
        raise ValueError("local_codes must be an int64 (n, columns) tensor")
    codes = codes.contiguous()
    n, ncols = codes.shape
    left0, right0, d = _flank_columns(ncols, k, model_type)
    keys = torch.empty(n, dtype=torch.int32, device=codes.device)
    status = torch.zeros(1, dtype=torch.int32, device=codes.device)
    with torch.cuda.device(codes.device):
        _lib.check(_lib.lib().mural_eval_kmer_keys(codes.data_ptr(), n, ncols, left0, right0, d, int(region_size), int(n_regions),
                                                  keys.data_ptr(), status.data_ptr(), _lib.current_stream_ptr(codes.device)))
    return keys, 5 ** (2 * d), status


def _group_table(keys, label, prob, n_groups, status):
    """float64 (n_groups, 1 + 2 n_class) numpy table: rows, rows per label, probability sums per class."""
    n, nc = prob.shape
    table = torch.zeros((n_groups, 1 + 2 * nc), dtype=torch.float64, device=prob.device)
    with torch.cuda.device(prob.device):
        _lib.check(_lib.lib().mural_eval_group_obs_pred(keys.data_ptr(), label.data_ptr(), prob.data_ptr(),
                                                       int(prob.dtype == torch.float64), n, nc, n_groups, table.data_ptr(),
                                                       status.data_ptr(), _lib.current_stream_ptr(prob.device)))
    return table.cpu().numpy()


def _pearson(a, b):
    """Pearson r of two float64 vectors the way ``Series.corr`` reports it (NaN for < 2 points or zero variance)."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    if a.shape[0] < 2:
        return float("nan")
    da, db = a - a.mean(), b - b.mean()
    den = np.sqrt((da * da).sum() * (db * db).sum())
    with np.errstate(invalid="ignore", divide="ignore"):
        return float((da * db).sum() / den) if den > 0 else float("nan")


def _group_means(table, nc, prob_dtype):
    """(observed rate, mean prob) per non-empty group; the mean takes the precision of the prob column, as a pandas
    group-by mean of a float32 column does."""
    live = table[:, 0] > 0
    cnt = table[live, 0:1]
    obs = table[live, 1:1 + nc] / cnt
    pred = table[live, 1 + nc:] / cnt
    if prob_dtype == torch.float32:
        pred = pred.astype(np.float32).astype(np.float64)
    return obs, pred


# ------------------------------------------------------------------------------------------------------------------
# k-mer and regional correlations
# ------------------------------------------------------------------------------------------------------------------
def freq_kmer_comp_multi(local_codes, mut_type, prob, k, n_class, model_type="snv"):
    """Per class, Pearson r between the observed frequency and the mean predicted probability over the groups of equal
    (k-1) flanking bases (evaluation.py:48-67).  Returns a list of n_class floats."""
    prob = _prob(prob)
    if prob.shape[1] != n_class:
        raise ValueError(f"prob has {prob.shape[1]} columns, n_class is {n_class}")
    keys, groups, status = _kmer_keys(local_codes, k, model_type)
    table = _group_table(keys, _label(mut_type, prob.shape[0]), prob, groups, status)
    _raise_status(status, "freq_kmer_comp_multi")
    obs, pred = _group_means(table, n_class, prob.dtype)
    return [_pearson(obs[:, c], pred[:, c]) for c in range(n_class)]


def corr_calc_sub(chrom_id, start, mut_type, prob, window, return_cv=False):
    """Per class, Pearson r between the observed and the predicted rate over ``window``-bp genomic windows
    (evaluation.py:124-193; 0 with fewer than 3 windows, as there).  ``chrom_id`` is an integer id per row (any
    numbering: the correlation does not depend on the window order).  With ``return_cv`` also the coefficients of
    variation the reference prints, as (corr, cv_obs, cv_pred)."""
    prob = _prob(prob)
    n, nc = prob.shape
    cid = _dev(chrom_id, "chrom_id").reshape(-1).to(torch.int32).contiguous()
    st = _dev(start, "start").reshape(-1).to(torch.int64).contiguous()
    if cid.shape[0] != n or st.shape[0] != n:
        raise ValueError("chrom_id / start / prob row counts differ")
    if n == 0:
        raise KeyError(0)                 # data.loc[0, 'chrom'] on an empty frame
    n_chrom = int(cid.max().item()) + 1
    # the same number of window slots on every chromosome, from the largest start overall (empty slots are dropped below)
    per = int(st.max().item()) // int(window) + 1
    base = np.arange(n_chrom, dtype=np.int64) * per
    n_groups = n_chrom * per
    if n_groups >= 1 << 27:
        raise ValueError(f"corr_calc_sub: {n_groups} window slots (window too small for these coordinates)")
    base_dev = torch.from_numpy(base).to(prob.device)
    keys = torch.empty(n, dtype=torch.int32, device=prob.device)
    status = torch.zeros(1, dtype=torch.int32, device=prob.device)
    with torch.cuda.device(prob.device):
        _lib.check(_lib.lib().mural_eval_window_keys(cid.data_ptr(), st.data_ptr(), n, int(window), base_dev.data_ptr(), n_chrom,
                                                    keys.data_ptr(), status.data_ptr(), _lib.current_stream_ptr(prob.device)))
    table = _group_table(keys, _label(mut_type, n), prob, n_groups, status)
    _raise_status(status, "corr_calc_sub")
    live = table[:, 0] > 0
    cnt = table[live, 0:1]
    obs, pred = table[live, 1:1 + nc] / cnt, table[live, 1 + nc:] / cnt
    corr, cv_o, cv_p = [], [], []
    for c in range(nc):
        with np.errstate(invalid="ignore", divide="ignore"):
            cv_o.append(float(np.std(obs[:, c], ddof=1) / np.mean(obs[:, c])) if obs.shape[0] > 1 else float("nan"))
            cv_p.append(float(np.std(pred[:, c], ddof=1) / np.mean(pred[:, c])) if obs.shape[0] > 1 else float("nan"))
        corr.append(_pearson(obs[:, c], pred[:, c]) if obs.shape[0] >= 3 else 0)
    return (corr, cv_o, cv_p) if return_cv else corr


def regional_score(local_codes, mut_type, prob, valid_size, kmer_list, n_class, model_type="snv"):
    """``Evaluator.evaluate_regional_score`` (evaluation.py:544-587): rows are cut into regions of 10000 rows (a tenth of
    the set when it has <= 100000 rows); score = sum over regions and over kmer_list[:2] of sum_c (1 - r_c)^2 with r the
    k-mer correlations inside the region; corr_list = per class, Pearson r across regions between the observed class
    frequency and the mean probability.  Returns (corr_list, score, n_regions)."""
    prob = _prob(prob)
    if valid_size > 10000 * 10:
        region_size = 10000
    else:
        region_size = valid_size // 10
    n_regions = valid_size // region_size           # ZeroDivisionError for valid_size < 10, like the reference
    label = _label(mut_type, prob.shape[0])
    score = 0.0
    region_tot = None
    for k in kmer_list[:2]:
        keys, groups, status = _kmer_keys(local_codes, k, model_type, region_size, n_regions)
        table = _group_table(keys, label, prob, n_regions * groups, status)
        _raise_status(status, "regional_score")
        table = table.reshape(n_regions, groups, -1)
        for r in range(n_regions):
            obs, pred = _group_means(table[r], n_class, prob.dtype)
            score += float(np.sum([(1 - _pearson(obs[:, c], pred[:, c])) ** 2 for c in range(n_class)]))
        region_tot = table.sum(axis=1)
    cnt = region_tot[:, 0:1]
    obs, pred = region_tot[:, 1:1 + n_class] / cnt, region_tot[:, 1 + n_class:] / cnt
    corr_list = [_pearson(obs[:, c], pred[:, c]) for c in range(n_class)]
    return corr_list, score, n_regions


# ------------------------------------------------------------------------------------------------------------------
# calibration metrics and the full-Dirichlet fit
# ------------------------------------------------------------------------------------------------------------------
def calibration_metrics(prob, label, n_bins=50):
    """NLL (mean cross entropy of log(prob) as logits), ECE, classwise ECE and Brier score the way ``calibrate_prob``
    reports them (evaluation.py:209-290, :340-358; ``n_bins`` bins (lower, upper] on a float32 ``linspace(0, 1)``).
    Returns a dict of floats."""
    prob = _prob(prob)
    n, nc = prob.shape
    lab = _label(label, n)
    bounds = torch.linspace(0, 1, n_bins + 1).to(prob.device)
    out = torch.zeros(2 + 3 * n_bins * (1 + nc), dtype=torch.float64, device=prob.device)
    status = torch.zeros(1, dtype=torch.int32, device=prob.device)
    with torch.cuda.device(prob.device):
        _lib.check(_lib.lib().mural_eval_calib_metrics(prob.data_ptr(), int(prob.dtype == torch.float64), lab.data_ptr(), n, nc,
                                                      n_bins, bounds.data_ptr(), out.data_ptr(), status.data_ptr(),
                                                      _lib.current_stream_ptr(prob.device)))
    n_seen = int(lab.max().item()) + 1 if n else 0          # ClasswiseECELoss: num_classes = max(labels) + 1
    _raise_status(status, "calibration_metrics")
    out = out.cpu().numpy()
    bins = out[2:].reshape(1 + nc, n_bins, 3)

    def ece_of(cells):
        live = cells[:, 0] > 0
        cnt = cells[live, 0]
        return float(np.sum(np.abs(cells[live, 1] / cnt - cells[live, 2] / cnt) * (cnt / n)))

    per_class = [ece_of(bins[1 + c]) for c in range(n_seen)]
    return {"nll": float(out[0] / n), "brier": float(out[1] / n), "ece": ece_of(bins[0]),
            "c_ece": float(np.mean(per_class)) if per_class else float("nan")}


def _fit_terms(prob, label, weights, need_hessian):
    n, k = prob.shape
    km = k * (k + 1)
    w = torch.from_numpy(np.ascontiguousarray(weights, dtype=np.float64)).to(prob.device)
    out = torch.zeros(1 + km + km * km, dtype=torch.float64, device=prob.device)
    status = torch.zeros(1, dtype=torch.int32, device=prob.device)
    with torch.cuda.device(prob.device):
        _lib.check(_lib.lib().mural_eval_dirichlet_fit_terms(prob.data_ptr(), int(prob.dtype == torch.float64), label.data_ptr(), n, k,
                                                            w.data_ptr(), int(need_hessian), out.data_ptr(), status.data_ptr(),
                                                            _lib.current_stream_ptr(prob.device)))
    _raise_status(status, "fit_full_dirichlet")
    out = out.cpu().numpy()
    return out[0] / n, out[1:1 + km] / n, out[1 + km:].reshape(km, km) / n


def fit_full_dirichlet(prob, label, reg_lambda=0.0, reg_mu=None, maxiter=1024, ftol=1e-12, gtol=1e-8):
    """Fit ``FullDirichletCalibrator(reg_lambda, reg_mu)`` (the reference's default 'FullDiri' is reg_lambda = 0) and return
    (weights (k, k + 1) float64 with the last row zero, final mean log-loss).  Same optimisation as
    dirichlet_python/dirichletcal/calib/multinomial.py:69-130, :246-327: identity start, Newton steps with the
    pseudo-inverse of the Hessian in the redundant ref_row parametrisation, the 41-point step-size ladder, the same
    stopping rules.  The loss / gradient / Hessian sums over the rows come from one kernel launch per evaluation (the
    reference differentiates the mean log-loss with jax); use the result with ``calibration.dirichlet_calibrate``."""
    prob = _prob(prob)
    n, k = prob.shape
    if k < 2 or k > _FIT_MAX_CLASSES:
        raise ValueError(f"fit_full_dirichlet supports 2..{_FIT_MAX_CLASSES} classes, got {k}")
    lab = _label(label, n)
    seen = torch.unique(lab).cpu().numpy()
    if seen.shape[0] != k or seen.min() != 0 or seen.max() != k - 1:
        raise ValueError("every class 0..n_class-1 must occur in the labels (the reference sizes the map by unique(y))")
    return newton_full_dirichlet(lambda w, need_hessian: _fit_terms(prob, lab, w, need_hessian), k, reg_lambda, reg_mu, maxiter,
                                 ftol, gtol)


# name -> (method, ref_row, reg_lambda, reg_mu, reg_norm): the calibrators `calibrate_prob` can be asked for (evaluation.py:303-316).
# They are ONE multinomial regression on [log p; 1] (multinomial.py) under different linear parametrisations of its (k, k + 1) weight
# matrix: 'Full' (every entry), 'Diag' (vector scaling: a diagonal and the intercepts), 'FixDiag' (temperature scaling: one scalar
# on the diagonal), with or without subtracting the last raw row, with L2 / ODIR regularisation of the effective weights.
CALIBRATORS = {
    "FullDiri": ("Full", True, 0.0, None, False),
    "FullDiriODIR": ("Full", True, 1e-2, 1e-2, False),
    "FullDiri1": ("Full", True, 0.0, None, True),
    "FullDiri2": ("Full", False, 0.0, None, False),
    "VectS": ("Diag", True, 0.0, None, False),
    "TempS": ("FixDiag", True, 0.0, None, False),
}


def _param_map(k, method, ref_row):
    """(identity start, M) with effective weights.ravel() = M @ params (multinomial.py:186-232: raw matrix of the method, minus its
    last row under ref_row)."""
    m = k + 1
    km = k * m
    if method == "Full":
        w0 = np.hstack([np.eye(k), np.zeros((k, 1))]).ravel()
        P = np.eye(km)
    elif method == "Diag":
        w0 = np.hstack([np.ones(k), np.zeros(k)])
        P = np.zeros((km, 2 * k))
        for j in range(k):
            P[j * m + j, j] = 1.0              # diagonal entry j
            P[j * m + k, k + j] = 1.0          # intercept j
    elif method == "FixDiag":
        w0 = np.ones(1)
        P = np.hstack([np.eye(k), np.zeros((k, 1))]).reshape(km, 1)
    else:
        raise ValueError(f"unknown calibration method {method}")
    if ref_row:                                # W = raw - raw[-1]   (multinomial.py:207-211)
        P = (np.eye(km) - np.kron(np.outer(np.ones(k), np.eye(k)[k - 1]), np.eye(m))) @ P
    return w0, P


def newton_full_dirichlet(row_terms, k, reg_lambda=0.0, reg_mu=None, maxiter=1024, ftol=1e-12, gtol=1e-8):
    """``newton_calibrator`` for the reference's default 'FullDiri' parametrisation."""
    return newton_calibrator(row_terms, k, "Full", True, reg_lambda, reg_mu, maxiter, ftol, gtol)


def newton_calibrator(row_terms, k, method="Full", ref_row=True, reg_lambda=0.0, reg_mu=None, maxiter=1024, ftol=1e-12, gtol=1e-8):
    """Host side of the fit (multinomial.py:246-327): ``row_terms(W (k, k+1), need_hessian) -> (mean loss, gradient (km,), Hessian
    (km, km))`` with respect to the EFFECTIVE weights supplies the data terms (the device kernel in ``fit_calibrator``); the
    method's parametrisation, the reference row and the regulariser are linear / quadratic in them and applied here."""
    import scipy.linalg
    m = k + 1
    km = k * m
    w0, M = _param_map(k, method, ref_row)
    offdiag = np.hstack([1.0 - np.eye(k), np.zeros((k, 1))]).ravel()
    icept = np.hstack([np.zeros((k, k)), np.ones((k, 1))]).ravel()
    scale = np.full(km, reg_lambda) if reg_mu is None else reg_lambda * offdiag + reg_mu * icept     # multinomial.py:159-168

    def terms(params, need_hessian):
        w = (M @ params).reshape(k, m)
        loss, g, h = row_terms(w, need_hessian)
        wv = w.ravel()
        loss += np.sum(scale * wv ** 2)
        g = g + 2.0 * scale * wv
        h = h + 2.0 * np.diag(scale)
        return float(loss), M.T @ g, M.T @ h @ M

    steps = np.hstack((np.linspace(1, 0.1, 10), np.logspace(-2, -32, 31)))
    weights = w0.copy()
    L_list = [terms(weights, False)[0]]
    for i in range(maxiter):
        _, gradient, hessian = terms(weights, True)
        if np.abs(gradient).sum() < gtol:
            break
        if method == "FixDiag":
            updates = gradient / hessian.ravel()              # multinomial.py:272-273
        else:
            try:
                updates = scipy.linalg.pinv(hessian) @ gradient
            except (np.linalg.LinAlgError, ValueError):
                updates = gradient
        for step_size in steps:
            tmp_w = weights - (updates * step_size).ravel()
            L = terms(tmp_w, False)[0]
            if (L - L_list[-1]) < 0:
                break
        L_list.append(float(L))
        if np.isnan(L):
            break
        if i >= 5:
            d = np.diff(L_list[-5:])
            if (float(np.min(d)) > -ftol) and (float(np.sum(d) > 0) == 0):
                weights = tmp_w.copy()
                break
        if (L_list[-1] - L_list[-2]) > 0:
            break
        weights = tmp_w.copy()
    final = terms(weights, False)[0]
    return (M @ weights).reshape(k, m), final


def fit_calibrator(prob, label, name="FullDiri", maxiter=1024, ftol=1e-12, gtol=1e-8):
    """Fit the calibrator ``calibrate_prob(..., calibr_name=name)`` builds (evaluation.py:303-316: 'FullDiri', 'FullDiriODIR',
    'FullDiri1', 'FullDiri2', 'VectS', 'TempS') and return (weights (k, k + 1) float64, final objective).  Every one of them is
    applied the same way afterwards: ``calibration.dirichlet_calibrate`` / ``mural_calibrate_rows`` with these weights."""
    if name not in CALIBRATORS:
        raise ValueError(f"unknown calibrator {name!r} (one of {sorted(CALIBRATORS)})")
    method, ref_row, reg_lambda, reg_mu, reg_norm = CALIBRATORS[name]
    prob = _prob(prob)
    n, k = prob.shape
    if k < 2 or k > _FIT_MAX_CLASSES:
        raise ValueError(f"fit_calibrator supports 2..{_FIT_MAX_CLASSES} classes, got {k}")
    lab = _label(label, n)
    seen = torch.unique(lab).cpu().numpy()
    if seen.shape[0] != k or seen.min() != 0 or seen.max() != k - 1:
        raise ValueError("every class 0..n_class-1 must occur in the labels (the reference sizes the map by unique(y))")
    if reg_norm:                                              # multinomial.py:81-86
        if reg_mu is None:
            reg_lambda = reg_lambda / (k * (k + 1))
        else:
            reg_lambda, reg_mu = reg_lambda / (k * (k - 1)), reg_mu / k
    return newton_calibrator(lambda w, need_hessian: _fit_terms(prob, lab, w, need_hessian), k, method, ref_row, reg_lambda, reg_mu,
                             maxiter, ftol, gtol)


# ------------------------------------------------------------------------------------------------------------------
# the reference's Evaluator, on tensors
# ------------------------------------------------------------------------------------------------------------------
class Evaluator:
    """Same reports as the reference's ``Evaluator`` (evaluation.py:489-587), from device tensors instead of DataFrames:
    ``local_codes`` / ``mut_type`` stand for ``data_local``'s us*/ds* and mut_type columns, ``y_prob`` for the probability
    frame.  Each ``evaluate_*`` prints through ``printer`` with the reference's labels and also returns the numbers."""

    _KMER = {"no_calibra": "mer correlation - all: ", "FullDiri": "mer correlation(after fdiri_cal)",
             "Poisson": "mer correlation(after Poisson_cal)"}
    _REGIONAL = {"no_calibra": "regional corr (validation):", "FullDiri": "regional corr (validation, after fdiri_cal):",
                 "Poisson": "regional corr (validation, after Poisson_cal):"}
    _CORR_LIST = {"no_calibra": "corr_list: ", "FullDiri": "corr_list(after fdiri_cal)", "Poisson": "corr_list(after Poisson_cal)"}
    _SCORE = {"no_calibra": "regional score: ", "FullDiri": "regional score(after fdiri_cal)",
              "Poisson": "regional score(after Poisson_cal)"}

    def __init__(self, local_codes, mut_type, y_prob, n_class, calibra="no_calibra", printer=print, model_type="snv"):
        if calibra not in self._KMER:
            raise KeyError(calibra)
        self.local_codes, self.mut_type, self.y_prob = local_codes, mut_type, _prob(y_prob)
        self.n_class, self.calibra, self.printer, self.model_type = n_class, calibra, printer, model_type
        self.metrics = {}

    def evaluate_kmer(self, kmer_list=(3, 5, 7)):
        out = {}
        for k in kmer_list:
            out[k] = freq_kmer_comp_multi(self.local_codes, self.mut_type, self.y_prob, k, self.n_class, self.model_type)
            self.printer(f"{k}{self._KMER[self.calibra]}", out[k])
        return out

    def evaluate_regional_corr(self, chrom_id, start, win_size_list=(100000, 500000)):
        out = {}
        for win_size in win_size_list:
            out[win_size] = corr_calc_sub(chrom_id, start, self.mut_type, self.y_prob, win_size)
            self.printer(self._REGIONAL[self.calibra], str(win_size) + "bp", out[win_size])
        return out

    def evaluate_regional_score(self, valid_size, kmer_list=(3, 5)):
        corr_list, score, n_regions = regional_score(self.local_codes, self.mut_type, self.y_prob, valid_size, list(kmer_list),
                                                     self.n_class, self.model_type)
        self.printer("n_regions:", n_regions)
        self.printer(self._CORR_LIST[self.calibra], corr_list)
        self.printer(self._SCORE[self.calibra], score, n_regions)
        self.metrics["score"] = score
        return corr_list, score, n_regions


def calibrate_prob(y_prob, y, printer=print, calibr_name="FullDiri"):
    """``calibrate_prob(y_prob, y, device, calibr_name)`` (evaluation.py:297-365): fit the named calibrator on the validation
    probabilities (the training loop asks for 'FullDiri'; 'FullDiriODIR', 'FullDiri1', 'FullDiri2', 'VectS', 'TempS' are the other
    choices), report NLL / ECE / CwECE / Brier before and after.  Returns (weights, nll after calibration, prob_cal (n, n_class)
    float64 on the device)."""
    y_prob = _prob(y_prob)
    weights, _ = fit_calibrator(y_prob, y, calibr_name)
    w = torch.from_numpy(weights).to(y_prob.device)
    tiny = torch.finfo(y_prob.dtype).tiny
    logp = torch.log(y_prob.clamp(tiny, 1 - tiny)).to(torch.float64)
    prob_cal = torch.softmax(logp @ w[:, :-1].T + w[:, -1], dim=1)
    before, after = calibration_metrics(y_prob, y), calibration_metrics(prob_cal, y)
    for tag, mtr in (("Before", before), ("After", after)):
        printer("%s %s scaling - NLL: %.8f, ECE: %.8f, CwECE: %.8f, Brier: %.8f" % (tag, calibr_name, mtr["nll"], mtr["ece"], mtr["c_ece"],
                                                                                   mtr["brier"]))
    return weights, after["nll"], prob_cal
