"""Build hygiene of the tower kernels (no GPU needed): they run at the 256-VGPR limit with two workgroups per CU, and a spill in a
hot loop costs several per cent of the headline (it happened by naming two kernel arguments in locals, by unrolling the layers with
their roles at compile time, by hoisted per-round lane offsets).  The Makefile keeps the compiler's resource reports of
csrc/snv_tower.hip and csrc/snv_tower_wave.hip; this test reads them."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REPORT = os.path.join(ROOT, "mural_amd", "csrc", "snv_tower.resources.txt")


REPORT_WAVE = os.path.join(ROOT, "mural_amd", "csrc", "snv_tower_wave.resources.txt")


def _kernels(report=REPORT):
    if not os.path.exists(report):
        pytest.skip("no compiler report (library built by an older Makefile)")
    text = open(report).read()
    out = {}
    for m in re.finditer(r"Function Name: (\S+)(.*?)(?=Function Name:|\Z)", text, re.S):
        body = m.group(2)
        vals = {k: int(v) for k, v in re.findall(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", body)}
        out[m.group(1)] = vals
    return out


@pytest.mark.parametrize("phase", [1, 2])
def test_stage_split_tower_kernels_do_not_spill(phase):
    ks = _kernels()
    name = [k for k in ks if "snv_towers_fused" in k and f"ILi{phase}E" in k]
    assert len(name) == 1, sorted(ks)
    r = ks[name[0]]
    assert r["VGPRs"] <= 256 and r["VGPRs Spill"] == 0 and r["ScratchSize"] == 0, r
    assert r["Occupancy"] >= 2, r


def test_wave_private_tower_kernels_stay_within_their_register_budget():
    """Every instance: no VGPR spill, no scratch, two waves per SIMD.  A spill here means a unit-invariant per-lane address (a hoisted
    64-bit lane pointer or LDS table address) came back: the guarded loads and stores go through range-checked buffer descriptors and
    the invariant lane indices are opaque for exactly that reason."""
    ks = _kernels(REPORT_WAVE)
    first = [k for k in ks if "snv_tower_wave" in k and "ILi1E" in k.split("snv_tower_wave")[1][:6]]
    short = [k for k in ks if "snv_tower_wave" in k and "ILi2E" in k.split("snv_tower_wave")[1][:6]]
    assert len(first) == 3 and len(short) == 2, sorted(ks)
    for k in first:
        r = ks[k]
        assert r["VGPRs"] <= 256 and r["VGPRs Spill"] == 0 and r["ScratchSize"] == 0 and r["Occupancy"] >= 2, (k, r)
    for k in short:
        r = ks[k]
        assert r["VGPRs"] <= 256 and r["VGPRs Spill"] == 0 and r["ScratchSize"] == 0 and r["Occupancy"] >= 2, (k, r)
