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


@pytest.mark.parametrize("phase,max_spill", [(0, 3), (3, 24)])
def test_small_call_tower_kernels_do_not_spill_more_than_recorded(phase, max_spill):
    """snv_towers_fused<0> / <3>: the launches of calls of up to 256 sites (the reference's default pred_batch_size is 16,
    MuRaL/commands/predict.py:90).  They sit at the 256-register limit of an eight-wave workgroup with 3 / 24 spilled registers (16 /
    92 bytes of scratch per lane) in a launch of ~36 us that is bound by its latencies, not its issue slots; the bound recorded here
    keeps a change from making it worse unnoticed (VERDICT r04 item 7 asked for zero: not reached)."""
    ks = _kernels()
    name = [k for k in ks if "snv_towers_fused" in k and f"ILi{phase}E" in k]
    assert len(name) == 1, sorted(ks)
    r = ks[name[0]]
    assert r["VGPRs"] <= 256 and r["VGPRs Spill"] <= max_spill and r["Occupancy"] >= 2, r


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


def _report(name):
    return os.path.join(ROOT, "mural_amd", "csrc", name + ".resources.txt")


def test_split_level0_convblock_instances_keep_their_occupancy():
    """convblock_kernel<8, TAIL, FRONT, MF = true> (conv1d.hip): no spill, no scratch, and the waves per SIMD its launch bound asks for
    (six for the instances with a front -- at seven they spill a register into scratch --, seven or eight otherwise): the
    fragments are requested behind the front / the staging exactly so that they are not live across them."""
    ks = _kernels(_report("conv1d"))
    split = [k for k in ks if "convblock_kernelILi8E" in k and k.split("convblock_kernelILi8E")[1].startswith(("Lb0ELb0ELb1E", "Lb0ELb1ELb1E", "Lb1ELb0ELb1E", "Lb1ELb1ELb1E"))]
    assert len(split) == 4, sorted(k for k in ks if "convblock_kernel" in k)
    for k in split:
        r = ks[k]
        need = 6 if k.split("convblock_kernelILi8E")[1][4:8] == "Lb1E" else 7      # (the instances with a front: six)
        assert r["VGPRs Spill"] == 0 and r["SGPRs Spill"] == 0 and r["ScratchSize"] == 0 and r["Occupancy"] >= need, (k, r)


def test_barrier_free_convs_and_blocks_do_not_spill():
    """conv1d_direct_kernel / conv1d_direct_poly_kernel (every instance that is launched) and convblock_direct_kernel<16> without spills
    at two waves per SIMD; the 24-channel block is allowed the two registers it spills today (114 weight registers)."""
    ks = _kernels(_report("conv1d_direct"))
    direct = [k for k in ks if "conv1d_direct" in k]
    assert len(direct) >= 16, sorted(ks)
    for k in direct:
        r = ks[k]
        assert r["VGPRs Spill"] == 0 and r["ScratchSize"] == 0 and r["Occupancy"] >= 2, (k, r)
    kb = _kernels(_report("convblock_mfma"))
    d16 = [k for k in kb if "convblock_direct_kernelILi16E" in k]
    d24 = [k for k in kb if "convblock_direct_kernelILi24E" in k]
    assert len(d16) == 1 and len(d24) == 1, sorted(kb)
    assert kb[d16[0]]["VGPRs Spill"] == 0 and kb[d16[0]]["ScratchSize"] == 0 and kb[d16[0]]["Occupancy"] >= 3, kb[d16[0]]
    assert kb[d24[0]]["VGPRs Spill"] <= 3 and kb[d24[0]]["Occupancy"] >= 2, kb[d24[0]]


def test_wave_private_training_convs_do_not_spill():
    """conv32w_fwd_kernel<NB> / conv32w_bwd_kernel<NB, FOLD> (csrc/conv32_wave.hip), every instance the step can launch: no VGPR
    spill, no scratch, two waves per SIMD.  VERDICT r04: every backward instance from six blocks up used to spill (59 registers,
    144 bytes of scratch per lane at nine blocks) -- a hoisted table of per-slot lane offsets whose reloads serialised the unit's
    loads, and a bias-gradient sum the optimiser sank to the loop latch, which kept all staged values alive through the MFMA phases."""
    ks = _kernels(_report("conv32_wave"))
    conv = [k for k in ks if "conv32w_fwd_kernel" in k or "conv32w_bwd_kernel" in k]
    assert len(conv) == 6 + 18, sorted(ks)      # forward x 6 block counts; backward x 6 x (plain | FOLD | FOLD with two residuals)
    for k in conv:
        r = ks[k]
        assert r["VGPRs"] <= 256 and r["VGPRs Spill"] == 0 and r["SGPRs Spill"] == 0 and r["ScratchSize"] == 0 and r["Occupancy"] >= 2, (k, r)


def test_persistent_level0_indel_kernels_do_not_spill():
    """indel_enc0_kernel<13 | 7> / indel_dec0_kernel (csrc/indel_level0.hip), the instances that are launched outside the phase-stamp
    diagnostics: no spill, no scratch, at least four waves per SIMD.  A scratch reload in their tile loop is behind a full `s_waitcnt vmcnt(0)`,
    i.e. behind the prefetch of the NEXT tile's input -- the one wait these kernels exist to remove (the first build had four spilled
    registers and exactly that wait in the middle of its matrix phase)."""
    ks = _kernels(_report("indel_level0"))
    run = [k for k in ks if ("indel_enc0_kernel" in k and k.split("indel_enc0_kernel")[1].startswith(("ILi13ELb0E", "ILi7ELb0E"))) or
           ("indel_dec0_kernel" in k and k.split("indel_dec0_kernel")[1].startswith("ILb0E"))]
    # encoder: 13 / 7 composed taps x (plain | also emitting the next level's strided conv) x (packed genome | symbol bytes); decoder
    assert len(run) == 9, sorted(ks)
    for k in run:
        r = ks[k]
        need = 4      # (both launches are bound by instruction issue: registers instead of re-derived addresses / LDS fragment reads)
        assert r["VGPRs Spill"] == 0 and r["SGPRs Spill"] == 0 and r["ScratchSize"] == 0 and r["Occupancy"] >= need, (k, r)


def test_short_row_block_keeps_fragments_and_prefetch_in_registers():
    """convblock_deep32_kernel<1..5> (csrc/convblock_deep.hip): 56 weight fragments, the next row's ten dwords, twelve skip values and
    five accumulators per lane -- no spill, no scratch at two workgroups per CU (at three the five-block instance spilled 11).  The
    two deepest levels' kernel keeps only the k = 5 fragments in registers (five / six waves share four SIMDs: 256 registers each)."""
    ks = _kernels(_report("convblock_deep"))
    deep = [k for k in ks if "convblock_deep32_kernel" in k or "convblock_tiny_kernel" in k]      # + the 40 x 16 and 48 x 8 levels
    assert len(deep) == 2 * 5 + 2, sorted(ks)      # (x 2: with / without the level's strided conv as the launch's front)
    for k in deep:
        r = ks[k]
        assert r["VGPRs Spill"] == 0 and r["SGPRs Spill"] == 0 and r["ScratchSize"] == 0 and r["Occupancy"] >= 2, (k, r)
