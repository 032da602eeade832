"""The suite runs on the debug flavour of the library (tests/conftest.py); these tests run the PRODUCT library -- what bench.py, smoke()
and every user load -- in a subprocess: same parity checks, development switches in the environment are ignored."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_flavours_differ_by_the_hooks_object_only():
    """Exported symbols: debug flavour = product library + the validation hooks (and the internal symbols of csrc/debug_hooks.hip)."""
    from mural_amd import _lib

    def exported(path):
        out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
        return {ln.split()[-1] for ln in out.splitlines() if ln.strip()}

    product, debug = exported(_lib.LIB_PATH), exported(_lib.DEBUG_LIB_PATH)
    assert product <= debug
    extra_c = {s for s in debug - product if not s.startswith("_Z") and not s.startswith("__hip_")}      # (__hip_cuid_*: one per object)
    assert extra_c == set(_lib.DEBUG_PROTOTYPES), extra_c ^ set(_lib.DEBUG_PROTOTYPES)
    assert not any("mural_debug" in s for s in product)


@pytest.mark.gpu
def test_product_library_passes_the_golden_checks_and_ignores_development_switches():
    env = {k: v for k, v in os.environ.items() if k != "MURAL_HIP_FLAVOR"}
    # switches that change results in the debug flavour (timing experiments, another kernel family): the product library must not see them
    env.update(MURAL_DEBUG_S1_ALIAS="1", MURAL_DEBUG_CW="8", MURAL_DEBUG_FIRST="3", MURAL_TRAIN_CONV_CL="1", MURAL_DEBUG_MLP="1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_product_checks.py")], capture_output=True, text=True, timeout=1200,
                         env=env, cwd=ROOT)
    assert out.returncode == 0 and "PRODUCT_LIBRARY_OK" in out.stdout, (out.stdout[-2000:], out.stderr[-4000:])
