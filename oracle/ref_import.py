"""Import the *reference* MuRaL package from /root/reference (build container only).

Test infrastructure.  The reference cannot travel to the GPU box; this module is
used only by ``oracle/make_golden.py`` (fixture generation) and by the optional
``-m "not gpu"`` cross-checks that skip when /root/reference is absent.

The reference's model file star-imports its evaluation module, which imports
prettytable / jax / dirichletcal; its preprocessing module imports pyBigWig /
pybedtools / Bio / h5py.  None of them take part in the hot-path arithmetic, so
they are replaced with inert placeholder modules (SURVEY.md section 8c recipe).
``MuRaL.model.nn_utils`` must be imported before ``model_snv`` (circular import).
"""
import importlib
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("MURAL_REFERENCE_ROOT", "/root/reference")


def available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "MuRaL"))


class _Anything:
    """Placeholder class: constructible, attribute access yields itself."""

    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return self

    def __getattr__(self, name):
        return _Anything()


def _placeholder(name):
    mod = types.ModuleType(name)
    mod.__path__ = []  # behave like a package so sub-imports resolve

    def _getattr(attr, _n=name):
        if attr.startswith("__"):
            raise AttributeError(attr)
        return _Anything

    mod.__getattr__ = _getattr
    return mod


_PLACEHOLDERS = [
    "prettytable", "jax", "jax.numpy", "jax.config", "dirichletcal", "dirichletcal.calib",
    "dirichletcal.calib.vectorscaling", "dirichletcal.calib.tempscaling",
    "dirichletcal.calib.fulldirichlet", "pyBigWig", "pybedtools", "Bio", "Bio.SeqIO",
    "h5py", "pynvml", "ray", "ray.tune",
]


def load():
    """Return a namespace with the reference modules used for fixture generation."""
    if not available():
        raise RuntimeError(f"reference tree not found at {REFERENCE_ROOT}")
    for name in _PLACEHOLDERS:
        if name not in sys.modules:
            try:
                importlib.import_module(name)
            except Exception:
                sys.modules[name] = _placeholder(name)
    jaxmod = sys.modules["jax"]
    if isinstance(getattr(jaxmod, "__getattr__", None), types.FunctionType):
        cfg = types.SimpleNamespace(update=lambda *a, **k: None)
        jaxmod.config = cfg
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    nn_utils = importlib.import_module("MuRaL.model.nn_utils")
    model_snv = importlib.import_module("MuRaL.model.model_snv")
    model_indel = importlib.import_module("MuRaL.model.model_indel")
    prep = importlib.import_module("MuRaL.data.preprocessing")
    return types.SimpleNamespace(nn_utils=nn_utils, model_snv=model_snv,
                                 model_indel=model_indel, preprocessing=prep)
