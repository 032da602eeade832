"""CPU restatement of the reference's validation-epoch analytics.  TEST INFRASTRUCTURE ONLY: imported by tests/ and by
oracle/make_golden.py, never by the product (mural_amd/evaluation.py runs these reductions in HIP kernels).

Follows MuRaL/evaluation/evaluation.py: freq_kmer_comp_multi (:48-67), corr_calc_sub (:124-193), calc_avg_prob (:195-203),
ECELoss (:209-231), ClasswiseECELoss (:233-270), BrierScore (:272-290), Evaluator.evaluate_regional_score (:544-587), and the
full-Dirichlet fit of dirichlet_python/dirichletcal/calib/multinomial.py (:69-130 fit, :153-172 objective, :246-327 Newton).

Pinned by tests/golden/analytics.npz (oracle/make_golden.py g13): the k-mer / regional / score / metric numbers there come
from the reference's own functions run in the build container.  The fit is only PARTLY pinned: jax is absent, so the golden
weights come from the reference's own Newton driver and objective run on numpy with THIS file's analytic gradient / Hessian
standing in for jax.grad / jax.hessian (checked against finite differences of the reference objective in make_golden).
"""
import numpy as np
import pandas as pd


def flank_columns(ncols, k, model_type):
    d = k // 2
    r = (ncols - 1) // 2 if model_type == "snv" else ncols // 2
    left = list(range(r - d, r))
    right0 = r + 1 if model_type == "snv" else r
    return left + list(range(right0, right0 + d))


def freq_kmer_comp_multi(codes, mut_type, prob, k, n_class, model_type="snv"):
    """evaluation.py:48-67 with the us*/ds* columns taken by position from the order-1 local encoding."""
    cols = flank_columns(codes.shape[1], k, model_type)
    names = ["c%d" % i for i in range(len(cols))]
    frame = pd.DataFrame(codes[:, cols], columns=names)
    out = []
    for i in range(n_class):
        f = pd.concat([frame, pd.DataFrame({"prob": prob[:, i]}), pd.DataFrame({"mut_type": np.asarray(mut_type) == i})], axis=1)
        f = f.groupby(names).mean()
        out.append(f["mut_type"].astype(float).corr(f["prob"].astype(float)))
    return out


def corr_calc_sub(chrom, start, mut_type, prob, window):
    """evaluation.py:124-193 on rows already sorted by (chrom, start): running sums per window (the probabilities are
    accumulated in their own dtype, row by row, as the reference's ``pred[j] += data.loc[i, name]`` does), then
    scipy's pearsonr over the windows (0 with fewer than 3)."""
    from scipy.stats import pearsonr
    n, nc = prob.shape
    rows = []
    obs, pred, count = [0] * nc, [0] * nc, 0
    last = (chrom[0], start[0] // window * window)
    for i in range(n):
        cur = (chrom[i], start[i] // window * window)
        if cur != last:
            rows.append([v / count for pair in zip(obs, pred) for v in pair])
            obs, pred, count, last = [0] * nc, [0] * nc, 0, cur
        obs[int(mut_type[i])] += 1
        for j in range(nc):
            pred[j] += prob[i, j]
        count += 1
    rows.append([v / count for pair in zip(obs, pred) for v in pair])
    res = np.asarray(rows, dtype=np.float64)
    out = []
    for j in range(nc):
        out.append(pearsonr(res[:, 2 * j], res[:, 2 * j + 1])[0] if res.shape[0] >= 3 else 0)
    return out


def regional_score(codes, mut_type, prob, valid_size, kmer_list, n_class, model_type="snv"):
    """evaluation.py:544-587 -> (corr_list, score, n_regions)."""
    region_size = 10000 if valid_size > 10000 * 10 else valid_size // 10
    n_regions = valid_size // region_size
    score = 0
    avg = []
    for i in range(n_regions):
        sl = slice(region_size * i, region_size * (i + 1))
        for k in kmer_list[:2]:
            corr = freq_kmer_comp_multi(codes[sl], mut_type[sl], prob[sl], k, n_class, model_type)
            score += np.sum([(1 - c) ** 2 for c in corr])
        lab = np.asarray(mut_type[sl])
        avg.append([np.sum(lab == c) / lab.shape[0] for c in range(n_class)] + [pd.Series(prob[sl, c]).mean() for c in range(n_class)])
    avg = pd.DataFrame(avg)
    return [avg[c].corr(avg[c + n_class]) for c in range(n_class)], float(score), n_regions


def calibration_metrics(prob, label, n_bins=50):
    """NLL / ECE / classwise ECE / Brier of calibrate_prob (evaluation.py:340-358) with log(prob) as logits."""
    import torch
    import torch.nn.functional as F
    logits = torch.log(torch.from_numpy(np.ascontiguousarray(prob)))
    labels = torch.from_numpy(np.asarray(label)).long()
    nll = F.cross_entropy(logits, labels, reduction="mean").item()
    sm = F.softmax(logits, dim=1)
    bounds = torch.linspace(0, 1, n_bins + 1)
    n = sm.shape[0]

    def binned(score, hit):
        tot = 0.0
        for lo, hi in zip(bounds[:-1], bounds[1:]):
            inb = score.gt(lo.item()) * score.le(hi.item())
            if inb.any():
                tot += abs(score[inb].mean().item() - hit[inb].float().mean().item()) * (inb.float().sum().item() / n)
        return tot

    conf, pred = torch.max(sm, 1)
    ece = binned(conf, pred.eq(labels))
    c_ece = float(np.mean([binned(sm[:, c], labels.eq(c)) for c in range(int(labels.max().item()) + 1)]))
    onehot = F.one_hot(labels, sm.shape[1]).to(sm.dtype)
    brier = (torch.sum((onehot - sm) ** 2) / n).item()
    return {"nll": nll, "ece": ece, "c_ece": c_ece, "brier": brier}


# ---- full-Dirichlet fit ------------------------------------------------------------------------------------------------
def fit_features(prob):
    """fulldirichlet.py:49-50: log(clip(X, tiny, 1 - tiny)) in X's dtype, then [X, 1] in float64."""
    tiny = np.finfo(prob.dtype).tiny
    x = np.log(np.clip(prob, tiny, 1 - tiny))
    return np.hstack((x.astype(np.float64), np.ones((len(x), 1))))


def effective_weights(params, k):
    raw = np.asarray(params, np.float64).reshape(-1, k + 1)
    return raw - raw[-1:, :]                               # multinomial.py:207-211 (ref_row)


def fit_row_terms(X_, label, w, need_hessian=True):
    """Mean loss, gradient and Hessian of multinomial.py:153-157 w.r.t. the effective weights w (k, k+1)."""
    n, m = X_.shape
    k = w.shape[0]
    z = X_ @ w.T
    z -= z.max(axis=1, keepdims=True)
    s = np.exp(z)
    s /= s.sum(axis=1, keepdims=True)
    eps = np.finfo(np.float64).eps
    sy = s[np.arange(n), label]
    live = (sy >= eps) & (sy <= 1 - eps)
    loss = float(np.mean(-np.log(np.clip(sy, eps, 1 - eps))))
    r = s.copy()
    r[np.arange(n), label] -= 1.0
    r[~live] = 0.0
    g = (r.T @ X_).ravel() / n
    h = np.zeros((k * m, k * m))
    if need_hessian:
        sl = s * live[:, None]
        for j in range(k):
            for j2 in range(k):
                a = (sl[:, j] if j == j2 else 0.0) - sl[:, j] * sl[:, j2]
                h[j * m:(j + 1) * m, j2 * m:(j2 + 1) * m] = (X_ * np.reshape(a, (-1, 1))).T @ X_ / n
    return loss, g, h


def fit_full_dirichlet(prob, label, maxiter=1024, ftol=1e-12, gtol=1e-8):
    """FullDirichletCalibrator().fit (reg_lambda = 0): multinomial.py:246-327 in the raw (redundant) parametrisation."""
    import scipy.linalg
    X_ = fit_features(prob)
    label = np.asarray(label).astype(np.int64)
    k = prob.shape[1]
    m = k + 1

    def raw_terms(params, need_hessian):
        w = effective_weights(params, k)
        loss, g, h = fit_row_terms(X_, label, w, need_hessian)
        g = g.reshape(k, m).copy()
        g[-1] -= g.sum(axis=0)                             # chain rule through W = raw - raw[-1]
        if need_hessian:
            h = h.reshape(k, m, k, m).copy()
            h[-1] -= h.sum(axis=0)
            h[:, :, -1] -= h.sum(axis=2)
            h = h.reshape(k * m, k * m)
        return loss, g.ravel(), h

    weights = np.hstack([np.eye(k), np.zeros((k, 1))]).ravel()
    L_list = [raw_terms(weights, False)[0]]
    for i in range(maxiter):
        _, gradient, hessian = raw_terms(weights, True)
        if np.abs(gradient).sum() < gtol:
            break
        updates = scipy.linalg.pinv(hessian) @ gradient
        for step_size in np.hstack((np.linspace(1, 0.1, 10), np.logspace(-2, -32, 31))):
            tmp_w = weights - updates * step_size
            L = raw_terms(tmp_w, False)[0]
            if (L - L_list[-1]) < 0:
                break
        L_list.append(float(L))
        if np.isnan(L):
            break
        if i >= 5:
            if (float(np.min(np.diff(L_list[-5:]))) > -ftol) & (float(np.sum(np.diff(L_list[-5:])) > 0) == 0):
                weights = tmp_w.copy()
                break
        if (L_list[-1] - L_list[-2]) > 0:
            break
        weights = tmp_w.copy()
    return effective_weights(weights, k), raw_terms(weights, False)[0]


# ------------------------------------------------------------------------------------------------------------------
# the other calibrators of calibrate_prob (evaluation.py:297-320): one multinomial regression (multinomial.py) under
# different linear parametrisations of the (k, k + 1) weight matrix, regularisers and with / without the reference row
# ------------------------------------------------------------------------------------------------------------------
CALIBRATORS = {
    # name: (method, ref_row, reg_lambda, reg_mu, reg_norm)
    "FullDiri": ("Full", True, 0.0, None, False),
    "FullDiriODIR": ("Full", True, 1e-2, 1e-2, False),      # evaluation.py:309-311
    "FullDiri1": ("Full", True, 0.0, None, True),           # reg_norm on reg_lambda = 0: the plain fit again
    "FullDiri2": ("Full", False, 0.0, None, False),
    "VectS": ("Diag", True, 0.0, None, False),              # vectorscaling.py:56-60 with logit_constant = 0: X = log p
    "TempS": ("FixDiag", True, 0.0, None, False),           # tempscaling.py:57-61
}


def raw_weights(params, k, method):
    """multinomial.py:186-205: the method's parameters as a raw (k, k + 1) matrix."""
    params = np.asarray(params, np.float64)
    if method == "Full":
        return params.reshape(k, k + 1)
    if method == "Diag":
        return np.hstack([np.diag(params[:k]), params[k:].reshape(-1, 1)])
    if method == "FixDiag":
        return np.hstack([np.eye(k) * params[0], np.zeros((k, 1))])
    raise ValueError(method)


def method_weights(params, k, method, ref_row):
    raw = raw_weights(params, k, method)
    return raw - raw[-1:, :] if ref_row else raw           # multinomial.py:207-211


def identity_params(k, method):
    """multinomial.py:216-232."""
    if method == "Full":
        return np.hstack([np.eye(k), np.zeros((k, 1))]).ravel()
    if method == "Diag":
        return np.hstack([np.ones(k), np.zeros(k)])
    return np.ones(1)


def reg_terms(w, k, reg_lambda, reg_mu):
    """multinomial.py:159-168 on the effective weights: value, gradient, (diagonal of the) Hessian."""
    wv = w.ravel()
    if reg_mu is None:
        scale = np.full(k * (k + 1), reg_lambda)
    else:
        scale = reg_lambda * np.hstack([1.0 - np.eye(k), np.zeros((k, 1))]).ravel() + reg_mu * np.hstack([np.zeros((k, k)), np.ones((k, 1))]).ravel()
    return float(np.sum(scale * wv ** 2)), 2.0 * scale * wv, 2.0 * scale


def fit_calibrator(prob, label, name, maxiter=1024, ftol=1e-12, gtol=1e-8):
    """calibrate_prob(..., calibr_name=name).fit: multinomial.py:69-130 + :246-327.  Returns (effective weights (k, k + 1), final
    objective)."""
    import scipy.linalg
    method, ref_row, reg_lambda, reg_mu, reg_norm = CALIBRATORS[name]
    X_ = fit_features(prob)
    label = np.asarray(label).astype(np.int64)
    k = prob.shape[1]
    km = k * (k + 1)
    if reg_norm:                                           # multinomial.py:81-86
        if reg_mu is None:
            reg_lambda = reg_lambda / (k * (k + 1))
        else:
            reg_lambda, reg_mu = reg_lambda / (k * (k - 1)), reg_mu / k
    w0 = identity_params(k, method)
    # the parametrisation is linear: column i of M is the effective weight matrix of the i-th unit parameter vector
    M = np.stack([method_weights(e, k, method, ref_row).ravel() for e in np.eye(w0.shape[0])], axis=1)

    def terms(params, need_hessian):
        w = (M @ params).reshape(k, k + 1)
        loss, g, h = fit_row_terms(X_, label, w, need_hessian)
        r, rg, rh = reg_terms(w, k, reg_lambda, reg_mu)
        g = g + rg
        if need_hessian:
            h = h + np.diag(rh)
        return loss + r, M.T @ g, (M.T @ h @ M if need_hessian else np.zeros((w0.shape[0],) * 2))

    weights = w0.copy()
    L_list = [terms(weights, False)[0]]
    for i in range(maxiter):
        _, gradient, hessian = terms(weights, True)
        if np.abs(gradient).sum() < gtol:
            break
        if method == "FixDiag":
            updates = gradient / hessian.ravel()           # multinomial.py:272-273
        else:
            updates = scipy.linalg.pinv(hessian) @ gradient
        for step_size in np.hstack((np.linspace(1, 0.1, 10), np.logspace(-2, -32, 31))):
            tmp_w = weights - (updates * step_size).ravel()
            L = terms(tmp_w, False)[0]
            if (L - L_list[-1]) < 0:
                break
        L_list.append(float(L))
        if np.isnan(L):
            break
        if i >= 5:
            if (float(np.min(np.diff(L_list[-5:]))) > -ftol) & (float(np.sum(np.diff(L_list[-5:])) > 0) == 0):
                weights = tmp_w.copy()
                break
        if (L_list[-1] - L_list[-2]) > 0:
            break
        weights = tmp_w.copy()
    assert km == M.shape[0]
    return (M @ weights).reshape(k, k + 1), terms(weights, False)[0]
