"""Pin the CPU oracle (oracle/) against vectors produced by the reference itself (tests/golden/).

CPU-only.  The reference ships no tests for this path; these fixtures were generated
by oracle/make_golden.py importing /root/reference in the build container.
"""
import glob
import os

import numpy as np
import pytest
import torch
import torch.nn as nn

from oracle import encode_ref
from tests import _util as U

SNV_FORWARD = sorted(os.path.basename(p) for p in glob.glob(os.path.join(U.GOLDEN, "snv_synth_*.npz"))
                     + glob.glob(os.path.join(U.GOLDEN, "snv_pretrained_*.npz")))
INDEL_FORWARD = sorted(os.path.basename(p) for p in glob.glob(os.path.join(U.GOLDEN, "indel_*.npz"))
                       if not os.path.basename(p).startswith("indel_train_"))


# ------------------------------------------------------------------ G1: encoders, bit exact
@pytest.mark.parametrize("model_type", ["snv", "indel"])
@pytest.mark.parametrize("neg", [False, True])
def test_kmer_encoder_bit_exact(model_type, neg):
    fx = U.load("encode.npz")
    codes = encode_ref.seq_to_codes(fx["seq"].tobytes().decode())
    sel = fx["strands"].astype(bool) == neg
    starts = fx["starts"][sel]
    strands = ["-" if neg else "+"] * len(starts)
    tag = "neg" if neg else "pos"
    for r, k in [(5, 3), (7, 3), (10, 3), (7, 1), (6, 2)]:
        want = fx[f"kmer_{model_type}_{tag}_r{r}_k{k}"]
        got = encode_ref.kmer_encode(codes, starts, strands, r, k, model_type)
        assert got.dtype == np.int64 and got.shape == want.shape
        assert np.array_equal(got, want), (model_type, tag, r, k)


@pytest.mark.parametrize("model_type", ["snv", "indel"])
@pytest.mark.parametrize("neg", [False, True])
def test_onehot_encoder_exact(model_type, neg):
    fx = U.load("encode.npz")
    codes = encode_ref.seq_to_codes(fx["seq"].tobytes().decode())
    sel = fx["strands"].astype(bool) == neg
    starts = fx["starts"][sel]
    strands = ["-" if neg else "+"] * len(starts)
    tag = "neg" if neg else "pos"
    w = np.arange(0)
    for R in (100, 1000):
        got = encode_ref.onehot_encode(codes, starts, strands, R, model_type)
        assert got.dtype == np.float32
        assert list(got.shape) == fx[f"ohesum_{model_type}_{tag}_R{R}"].tolist()
        w = (np.arange(got.shape[2], dtype=np.float64) % 97 + 1.0)
        chk = (got.astype(np.float64) * w[None, None, :]).sum(axis=2)
        assert np.array_equal(chk, fx[f"ohechk_{model_type}_{tag}_R{R}"])
        if R == 100:
            assert np.array_equal(got, fx[f"ohe_{model_type}_{tag}_R{R}"])


def test_pack_roundtrip():
    fx = U.load("encode.npz")
    codes = encode_ref.seq_to_codes(fx["seq"].tobytes().decode())
    packed, mask = encode_ref.pack_codes(codes)
    back = encode_ref.unpack_codes(packed, mask, len(codes))
    assert np.array_equal(back, np.where(codes >= 4, 4, codes))


# ------------------------------------------------------------------ G3-G5: SNV forward
@pytest.mark.parametrize("name", SNV_FORWARD)
def test_snv_forward_matches_reference(name):
    fx = U.load(name)
    model = U.snv_oracle_from_hp(fx["hp"])
    model.load_state_dict(U.snv_state_for(fx, model))
    model.eval()
    cat = torch.from_numpy(fx["cat"])
    x = U.onehot(fx["codes"])
    with torch.no_grad():
        out = model((torch.zeros(len(cat), 1, dtype=torch.float64), cat), x).numpy()
    want = fx["logp"] if "logp" in fx.files else fx["out"]
    assert np.abs(out - want).max() <= 2e-6, name


def test_snv_state_dict_layout():
    fx = U.load("snv_pretrained_human_AT.npz")
    model = U.snv_oracle_from_hp(fx["hp"])
    sd = model.state_dict()
    assert len(sd) == 302
    assert sum(v.numel() for v in sd.values()) == 140533
    assert sum(p.numel() for p in model.parameters()) == 86904
    keys = list(sd.keys())
    assert keys[0] == "emb_layer.weight" and keys[-1] == "local_fc.0.bias"
    assert sd["first_bn_layer.weight"].shape == (0,)
    assert "RBs1.0.layer.2.weight" in sd and "RBs1_2.1.layer.5.bias" in sd


def test_snv_taps():
    fx = U.load("snv_taps.npz")
    model = U.snv_oracle_from_hp(fx["hp"])
    model.load_state_dict(U.snv_state_for(fx, model))
    model.eval()
    taps = {}
    cat = torch.from_numpy(fx["cat"])
    with torch.no_grad():
        out = model((torch.zeros(len(cat), 1, dtype=torch.float64), cat), U.onehot(fx["codes"]), taps=taps)
    assert np.abs(out.numpy() - fx["out"]).max() <= 2e-6
    pairs = {"maxpool1": "pool1", "maxpool2": "pool2", "conv2": "conv2", "maxpool3": "pool3", "conv3": "conv3",
             "distal_fc1": "fc"}
    for ref_name, mine in pairs.items():
        assert np.abs(taps[mine].numpy() - fx["tap::" + ref_name]).max() <= 1e-5, ref_name
        assert np.abs(taps[mine + "_2"].numpy() - fx["tap::" + ref_name.replace("fc1", "fc2") + ("" if "fc" in ref_name else "_2")]).max() <= 1e-5


# ------------------------------------------------------------------ G7: one training step
@pytest.mark.parametrize("tag", ["T", "S"])
def test_snv_train_step(tag):
    fx = U.load(f"snv_train_{tag}.npz")
    # the fixture is clear of max-pool near-ties (oracle/make_golden.py: pool_margins), so its gradients do not hinge on float32
    # summation order
    assert float(fx["pool_margin"]) >= 8e-6
    model = U.snv_oracle_from_hp(fx["hp"], drops=(0.0, 0.0, 0.0))
    model.load_state_dict(U.snv_state_for(fx, model))
    model.train()
    cat = torch.from_numpy(fx["cat"])
    preds = model((torch.zeros(len(cat), 1, dtype=torch.float64), cat), U.onehot(fx["codes"]))
    loss = nn.CrossEntropyLoss(reduction="sum")(preds, torch.from_numpy(fx["y"]))
    loss.backward()
    assert abs(loss.item() - float(fx["loss"])) <= 1e-4 * abs(float(fx["loss"]))
    for k, p in model.named_parameters():
        if ".layer." in k or p.numel() == 0:
            continue
        want = fx["g::" + k]
        scale = max(np.abs(want).max(), 1e-6)
        assert np.abs(p.grad.numpy() - want).max() <= 1e-4 * scale + 1e-6, k
    for k, b in model.named_buffers():
        if ".layer." in k or k.endswith("num_batches_tracked") or b.numel() == 0:
            continue
        assert np.abs(b.numpy() - fx["b::" + k]).max() <= 1e-5, k


# ------------------------------------------------------------------ G14: one INDEL training step
@pytest.mark.parametrize("tag", ["rev", "norev"])
def test_indel_train_step(tag):
    fx = U.load(f"indel_train_{tag}.npz")
    model = U.indel_oracle_from_hp(fx["hp"], fx["down"])
    model.load_state_dict(U.indel_state_for(fx, model))
    model.train()
    model.out_fc[1].p = 0.0
    preds = model(U.onehot(fx["codes"]))
    loss = nn.CrossEntropyLoss(reduction="sum")(preds, torch.from_numpy(fx["y"]))
    loss.backward()
    assert np.abs(preds.detach().numpy() - fx["preds"]).max() <= 1e-5
    assert abs(loss.item() - float(fx["loss"])) <= 1e-5 * abs(float(fx["loss"]))
    for k, p in model.named_parameters():
        want = fx["g::" + k]
        assert np.abs(p.grad.numpy() - want).max() <= 1e-4 * (np.abs(want).max() + 1e-2), k
    for k, b in model.named_buffers():
        assert np.abs(b.numpy().astype(np.float64) - fx["b::" + k]).max() <= 1e-5, k


# ------------------------------------------------------------------ G8: INDEL forward
@pytest.mark.parametrize("name", INDEL_FORWARD)
def test_indel_forward_matches_reference(name):
    fx = U.load(name)
    model = U.indel_oracle_from_hp(fx["hp"], fx["down"])
    model.load_state_dict(U.indel_state_for(fx, model))
    model.eval()
    with torch.no_grad():
        out = model(U.onehot(fx["codes"])).numpy()
    assert np.abs(out - fx["out"]).max() <= 1e-5 * max(1.0, np.abs(fx["out"]).max()), name
