import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# The suite runs on the DEBUG flavour of the library (mural_amd/libmural_hip_debug.so): the very objects of libmural_hip.so plus
# csrc/debug_hooks.hip -- the validation hooks of include/mural_hip_debug.h (LDS poisoning in front of every GPU test, the conv kernels
# on their own, workspace guard zones) and the development switches the A/B tests flip.  The product library exports no hook and reads
# no switch; tests/test_product_library.py checks it (exports, a stray switch is ignored, smoke + golden parity through it).
os.environ.setdefault("MURAL_HIP_FLAVOR", "debug")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(autouse=True)
def _poison_lds_before_gpu_tests(request):
    """Every `-m gpu` test starts from LDS full of NaN on every CU (mural_debug_poison_lds): a kernel whose result depends on LDS it
    has not written meets NaN instead of the previous kernel's leftovers and fails its parity check (this is how the split ConvBlock's
    two unwritten tile entries showed up reproducibly instead of once in two runs)."""
    if request.node.get_closest_marker("gpu") is not None:
        import torch
        if torch.cuda.is_available():
            from mural_amd import _lib
            _lib.check(_lib.lib().mural_debug_poison_lds(_lib.current_stream_ptr(torch.device("cuda", 0))))
    yield
