"""Post-head probability calibration of the prediction path (MuRaL/scripts/run_predict.py:217-225).

  * ``dirichlet_calibrate``  -- the fitted full-Dirichlet map of ``model.fdiri_cal.pkl``:
        softmax(W . [log(clip(p, tiny, 1 - tiny)); 1])            W: (n_class, n_class + 1), float64
    (dirichlet_python/dirichletcal/calib/fulldirichlet.py:78-80, calib/multinomial.py:60-64, :235-244, utils.py:5-7).
  * ``load_dirichlet_weights`` -- reads W out of the reference's pickles WITHOUT importing dirichletcal / jax: a restricted
    unpickler maps the two calibrator classes to inert holders and ``jax._src.array._reconstruct_array`` (present in the
    INDEL pickles) to its numpy equivalent; any other global is refused.
  * ``calibrate_device``     -- the whole chain (softmax -> Dirichlet map -> Poisson -> mu scaling) in one HIP kernel on the
    (n, n_class) device tensor the model returned; the numpy functions here are the host-side counterparts for arrays that
    already left the device.
  * ``poisson_calibrate`` lives in ``mural_amd.data.ingest`` (MuRaL/model/calibration.py:10-23).
  * ``mu_scaling_factor`` / ``apply_scaling`` -- the per-generation rate scaling of MuRaL/scripts/scaling.py:10-28, :76-93.
"""
import pickle

import numpy as np


class _Holder:
    """Stands in for FullDirichletCalibrator / MultinomialRegression: keeps the pickled attribute dict."""

    def __setstate__(self, state):
        self.__dict__.update(state)


def _reconstruct_jax_array(fun, args, arr_state, aval_state):
    arr = fun(*args)
    arr.__setstate__(arr_state)
    return arr


_ALLOWED = {
    ("dirichletcal.calib.fulldirichlet", "FullDirichletCalibrator"): _Holder,
    ("dirichletcal.calib.multinomial", "MultinomialRegression"): _Holder,
    ("jax._src.array", "_reconstruct_array"): _reconstruct_jax_array,
}
_NUMPY_OK = {("numpy", "ndarray"), ("numpy", "dtype"), ("numpy.core.multiarray", "_reconstruct"),
             ("numpy._core.multiarray", "_reconstruct"), ("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar")}


class _Restricted(pickle.Unpickler):
    def find_class(self, module, name):
        if (module, name) in _ALLOWED:
            return _ALLOWED[(module, name)]
        if (module, name) in _NUMPY_OK:
            return super().find_class(module, name)
        raise pickle.UnpicklingError(f"calibrator pickle references {module}.{name}: refused")


def load_dirichlet_weights(path):
    """(n_class, n_class + 1) float64 weight matrix of a ``model.fdiri_cal.pkl`` written by the reference."""
    with open(path, "rb") as fh:
        obj = _Restricted(fh).load()
    inner = getattr(obj, "calibrator_", None)
    w = getattr(inner, "weights_", None)
    if w is None:
        raise ValueError(f"{path}: no fitted FullDirichletCalibrator inside (calibrator_.weights_ missing)")
    w = np.asarray(w, dtype=np.float64)
    if w.ndim != 2 or w.shape[1] != w.shape[0] + 1:
        raise ValueError(f"{path}: unexpected weight shape {w.shape}")
    return w


def dirichlet_calibrate(prob, weights):
    """Apply the full-Dirichlet calibration map to an (n, n_class) probability array; returns float64."""
    prob = np.asarray(prob)
    eps = np.finfo(prob.dtype).tiny                                    # clip_for_log uses the INPUT's dtype
    s = np.log(np.clip(prob, eps, 1 - eps))
    s1 = np.hstack((s, np.ones((len(s), 1))))
    z = np.dot(s1, np.asarray(weights, dtype=np.float64).transpose())
    z = z - np.max(z, axis=1).reshape(-1, 1)
    e = np.exp(z)
    return e / np.sum(e, axis=1).reshape(-1, 1)


def calibrate_device(output, dirichlet_weights=None, poisson=False, scale_factor=None, input_is_prob=False, dtype=None):
    """The post-head chain of run_predict.py:214-225 on the device in ONE kernel (csrc/calibrate.hip): softmax of the model
    output (unless `input_is_prob`), the full-Dirichlet map, Poisson calibration and mu scaling, each optional.  `output`:
    (n, n_class) float32 tensor on a HIP device.  Returns a device tensor (float64 by default, like the host functions)."""
    import ctypes as C

    import torch

    from . import _lib
    x = _lib.require_cuda(output, "output").to(torch.float32).contiguous()
    if x.dim() != 2:
        raise ValueError(f"output must be (n, n_class), got {tuple(x.shape)}")
    n, k = x.shape
    dtype = torch.float64 if dtype is None else dtype
    if dtype not in (torch.float64, torch.float32):
        raise ValueError("dtype must be float64 or float32")
    w = None
    if dirichlet_weights is not None:
        wh = np.ascontiguousarray(dirichlet_weights, dtype=np.float64)
        if wh.shape != (k, k + 1):
            raise ValueError(f"dirichlet_weights must be ({k}, {k + 1}), got {wh.shape}")
        w = torch.from_numpy(wh).to(x.device)
    out = torch.empty((n, k), dtype=dtype, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().mural_calibrate_rows(x.data_ptr(), n, k, int(bool(input_is_prob)), w.data_ptr() if w is not None else None,
                                                  int(bool(poisson)), float(scale_factor or 0.0), out.data_ptr(),
                                                  int(dtype == torch.float64), _lib.current_stream_ptr(x.device)))
    return out


def mu_scaling_factor(prob, genomewide_mu, m_proportion, g_proportion=1.0):
    """Factor that turns relative mutation probabilities into per-generation rates (MuRaL/scripts/scaling.py:76-93):
    genomewide_mu * n_sites * m_proportion / g_proportion / sum over sites of (prob1 + ... + prob_{k-1}); `prob` holds the sites
    used as the benchmark (all predicted sites, or those inside the benchmark regions)."""
    prob = np.asarray(prob, dtype=np.float64)
    return float(genomewide_mu * prob.shape[0] * m_proportion / g_proportion / prob[:, 1:].sum())


def apply_scaling(prob, scale_factor):
    """scripts/scaling.py:10-28: the mutation classes are multiplied by the factor, class 0 becomes 1 - their sum."""
    out = np.asarray(prob, dtype=np.float64).copy()
    out[:, 1:] *= scale_factor
    out[:, 0] = 1.0 - out[:, 1:].sum(axis=1)
    return out


def save_dirichlet_calibrator(weights, path):
    """Write a ``model.fdiri_cal.pkl`` the reference can load (MuRaL/training.py:574-575 pickles its fitted
    ``FullDirichletCalibrator``; scripts/run_predict.py then calls ``predict_proba`` on it): the same two objects with the same
    attribute dictionaries, pickled by class reference without importing dirichletcal / jax here.  `weights`: the (k, k + 1)
    float64 matrix of ``evaluation.fit_full_dirichlet`` / ``load_dirichlet_weights``."""
    import sys
    import types
    w = np.ascontiguousarray(weights, dtype=np.float64)
    if w.ndim != 2 or w.shape[1] != w.shape[0] + 1:
        raise ValueError(f"weights must be (k, k + 1), got {w.shape}")
    k = w.shape[0]
    names = {"dirichletcal.calib.fulldirichlet": "FullDirichletCalibrator", "dirichletcal.calib.multinomial": "MultinomialRegression"}
    saved = {m: sys.modules.get(m) for m in list(names) + ["dirichletcal", "dirichletcal.calib"]}
    classes = {}
    try:
        for mod, cls in names.items():            # look-alike modules so that pickle records the reference's global names
            m = types.ModuleType(mod)
            c = type(cls, (), {})
            c.__module__ = mod
            setattr(m, cls, c)
            sys.modules[mod] = m
            classes[cls] = c
        for pkg in ("dirichletcal", "dirichletcal.calib"):
            sys.modules[pkg] = types.ModuleType(pkg)
        common = dict(reg_lambda=0.0, reg_mu=None, initializer="identity", reg_norm=False, ref_row=True, optimizer="auto")
        inner = classes["MultinomialRegression"]()
        inner.__dict__.update(dict(weights_0=None, method="Full", reg_format=None, classes=np.arange(k, dtype=np.int64), weights_=w,
                                   weights_0_=np.hstack([np.eye(k), np.zeros((k, 1))]).ravel(), **common))
        outer = classes["FullDirichletCalibrator"]()
        outer.__dict__.update(dict(weights_init=None, weights_=None, calibrator_=inner, **common))
        with open(path, "wb") as fh:
            pickle.dump(outer, fh, protocol=4)
    finally:
        for m, old in saved.items():
            if old is None:
                sys.modules.pop(m, None)
            else:
                sys.modules[m] = old
