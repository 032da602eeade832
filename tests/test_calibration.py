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
