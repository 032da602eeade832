"""Dirichlet calibration (mural_amd/calibration.py) vs outputs of the reference's own FullDirichletCalibrator on two shipped
calibrators (tests/golden/dirichlet.npz, oracle/make_golden.py g12), and the jax-free pickle reader."""
import pickle
import sys
import types

import numpy as np
import pytest

from mural_amd import calibration as K
from tests import _util as U


@pytest.mark.parametrize("tag", ["snv", "indel"])
def test_dirichlet_calibrate_matches_reference(tag):
    fx = U.load("dirichlet.npz")
    got = K.dirichlet_calibrate(fx[tag + "_prob"], fx[tag + "_w"])
    assert got.dtype == np.float64 and got.shape == fx[tag + "_cal"].shape
    assert np.abs(got - fx[tag + "_cal"]).max() <= 1e-14
    assert np.allclose(got.sum(axis=1), 1.0, atol=1e-12)


class _Fake:
    pass


def _jax_like(fun, args, arr_state, aval_state):      # pickled by reference: looked up as jax._src.array._reconstruct_array
    raise AssertionError("the reader must not import jax")


class _JaxArray:
    """Pickles like a jax Array: reduce -> (jax._src.array._reconstruct_array, (fun, args, arr_state, aval_state))."""

    def __init__(self, a):
        self.a = a

    def __reduce__(self):
        fun, args, state = self.a.__reduce__()
        return _jax_like, (fun, args, state, {"weak_type": False})


def test_load_dirichlet_weights_without_dirichletcal_or_jax(tmp_path):
    w = np.random.default_rng(3).normal(size=(4, 5))
    names = {"dirichletcal": None, "dirichletcal.calib": None, "dirichletcal.calib.fulldirichlet": "FullDirichletCalibrator",
             "dirichletcal.calib.multinomial": "MultinomialRegression", "jax": None, "jax._src": None,
             "jax._src.array": "_reconstruct_array"}
    classes = {}
    try:
        for mod, attr in names.items():     # temporary look-alike modules so that pickle writes the reference's global names
            m = types.ModuleType(mod)
            m.__path__ = []
            sys.modules[mod] = m
            if attr == "_reconstruct_array":
                _jax_like.__module__, _jax_like.__qualname__, _jax_like.__name__ = mod, attr, attr
                setattr(m, attr, _jax_like)
            elif attr:
                cls = type(attr, (_Fake,), {"__module__": mod})
                setattr(m, attr, cls)
                classes[attr] = cls
        for variant, arr in (("numpy", w), ("jax", _JaxArray(w))):
            inner = classes["MultinomialRegression"]()
            inner.weights_ = arr
            inner.method = "Full"
            outer = classes["FullDirichletCalibrator"]()
            outer.calibrator_ = inner
            outer.reg_lambda = 0.0
            with open(tmp_path / f"{variant}.pkl", "wb") as fh:
                pickle.dump(outer, fh)
    finally:
        for mod in names:
            sys.modules.pop(mod, None)
    for variant in ("numpy", "jax"):
        got = K.load_dirichlet_weights(tmp_path / f"{variant}.pkl")
        assert got.dtype == np.float64 and np.array_equal(got, w)
    evil = tmp_path / "evil.pkl"
    with open(evil, "wb") as fh:
        pickle.dump(print, fh)
    with pytest.raises(pickle.UnpicklingError, match="refused"):
        K.load_dirichlet_weights(evil)


def test_saved_calibrator_round_trips_and_loads_in_the_reference(tmp_path):
    """save_dirichlet_calibrator -> our restricted reader; and, when the reference tree is mounted, -> the reference's own
    classes (numpy standing in for jax, as in oracle/make_golden.py g12), whose predict_proba equals dirichlet_calibrate."""
    import os
    rng = np.random.default_rng(4)
    k = 4
    w = np.vstack([rng.normal(size=(k - 1, k + 1)), np.zeros((1, k + 1))])
    path = str(tmp_path / "model.fdiri_cal.pkl")
    K.save_dirichlet_calibrator(w, path)
    assert np.array_equal(K.load_dirichlet_weights(path), w)
    assert "dirichletcal.calib.fulldirichlet" not in sys.modules          # the look-alike modules are gone again
    from oracle import ref_import
    root = os.path.join(ref_import.REFERENCE_ROOT, "dirichlet_python")
    if not os.path.isdir(root):
        return
    saved = {m: sys.modules.get(m) for m in list(sys.modules) if m == "jax" or m.startswith("jax.") or m.startswith("dirichletcal")
             or m == "autograd" or m.startswith("autograd.")}
    for m in saved:
        sys.modules.pop(m, None)

    def mod(name, **kw):
        mm = types.ModuleType(name)
        mm.__path__ = []
        for a, v in kw.items():
            setattr(mm, a, v)
        sys.modules[name] = mm
        return mm

    noop = lambda *a, **kw: (lambda *b, **kb: None)      # noqa: E731
    jax = mod("jax", numpy=np, grad=noop, hessian=noop)
    jax.config = mod("jax.config", config=types.SimpleNamespace(update=lambda *a, **kw: None)).config
    sys.modules["jax.numpy"] = np
    mod("autograd", grad=noop, hessian=noop, numpy=np)
    sys.modules["autograd.numpy"] = np
    sys.path.insert(0, root)
    try:
        __import__("dirichletcal")
        with open(path, "rb") as fh:
            cal = pickle.load(fh)
        assert type(cal).__module__ == "dirichletcal.calib.fulldirichlet"
        prob = rng.dirichlet([20, 1, 1, 1], size=32).astype(np.float32)
        assert np.abs(np.asarray(cal.predict_proba(prob)) - K.dirichlet_calibrate(prob, w)).max() <= 1e-14
    finally:
        sys.path.pop(0)
        for m in [m for m in sys.modules if m == "jax" or m.startswith("jax.") or m.startswith("dirichletcal") or m == "autograd"
                  or m.startswith("autograd.")]:
            sys.modules.pop(m, None)
        sys.modules.update({m: v for m, v in saved.items() if v is not None})
