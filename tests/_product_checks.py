"""Runs IN A SUBPROCESS of tests/test_product_library.py with MURAL_HIP_FLAVOR unset and stray development switches in the
environment: the PRODUCT library (mural_amd/libmural_hip.so -- no validation hook, no development switch) through the same golden
parity checks the suite makes on the debug flavour: the driver's smoke(), every SNV forward fixture (the three shipped human
checkpoints included), the INDEL forward fixtures, one training step against the reference's gradients (G7 T and S through the
composition bench.py times) and the config-1 file-to-table chain."""
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    assert os.environ.get("MURAL_HIP_FLAVOR", "") != "debug"
    from mural_amd import _lib
    assert _lib.flavor() == "product"
    lib = _lib.lib()
    assert lib._name.endswith("libmural_hip.so") and not hasattr(lib, "mural_debug_poison_lds")
    import __graft_entry__ as g
    g.smoke()
    import pathlib
    from tests import test_gpu_config5, test_gpu_indel, test_gpu_snv, test_gpu_train
    n = 0
    for name in test_gpu_snv.SNV_FORWARD:
        test_gpu_snv.test_forward_dense_matches_reference(name)
        n += 1
    for name in test_gpu_indel.INDEL_FORWARD:
        test_gpu_indel.test_forward_matches_reference(name)
        n += 1
    for tag in ("T", "S"):
        test_gpu_train.test_train_step_through_the_bench_composition_matches_reference(tag)
        n += 1
    with tempfile.TemporaryDirectory() as tmp:
        test_gpu_config5.test_config1_example_files_to_calibrated_table(pathlib.Path(tmp), False)
        n += 1
    print("PRODUCT_LIBRARY_OK %d checks" % n)


if __name__ == "__main__":
    main()
