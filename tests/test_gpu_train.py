"""GPU parity of the training step (forward with batch-stat BN, CE-sum loss, backward) against the reference's golden
vectors: loss, every parameter gradient (relative 1e-4 of the tensor's max) and BN running statistics after the step."""
import os

import numpy as np
import pytest
import torch
import torch.nn as nn

from tests import _util as U
from tests.test_gpu_snv import product_from_hp

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("conv", ["wave", "tile"])
@pytest.mark.parametrize("tag", ["T", "S"])
def test_train_step_matches_reference(tag, conv, monkeypatch):
    # both conv kernel families of the composed step: wave-private units (csrc/conv32_wave.hip, the default) and workgroup tiles
    # (csrc/conv32_cl.hip, MURAL_TRAIN_CONV_CL=1)
    if conv == "tile":
        monkeypatch.setenv("MURAL_TRAIN_CONV_CL", "1")
    fx = U.load(f"snv_train_{tag}.npz")
    model, _ = product_from_hp(fx["hp"])
    orc = U.snv_oracle_from_hp(fx["hp"], drops=(0.0, 0.0, 0.0))
    model.load_state_dict(U.snv_state_for(fx, orc))
    for m in model.modules():            # the fixture was generated with every dropout rate at 0
        if isinstance(m, nn.Dropout):
            m.p = 0.0
    model = model.cuda().train()
    cat = torch.from_numpy(fx["cat"]).cuda()
    x = U.onehot(fx["codes"]).cuda()
    preds = model((torch.zeros(len(cat), 1, dtype=torch.float64, device="cuda"), cat), x)
    assert np.abs(preds.detach().cpu().numpy() - fx["preds"]).max() <= 2e-4
    loss = nn.CrossEntropyLoss(reduction="sum")(preds, torch.from_numpy(fx["y"]).cuda())
    model.zero_grad()
    loss.backward()
    assert abs(loss.item() - float(fx["loss"])) <= 1e-4 * abs(float(fx["loss"]))
    worst = ("", 0.0)
    for k, p in model.named_parameters():
        if ".layer." in k or p.numel() == 0:
            continue
        want = fx["g::" + k]
        assert p.grad is not None, k
        # biases in front of a batch-statistics BN have a mathematically zero gradient (1e-7 rounding noise in the
        # reference too): the absolute floor keeps those from being compared relative to noise
        scale = float(np.abs(want).max()) + 1e-2
        err = float(np.abs(p.grad.cpu().numpy() - want).max()) / scale
        if err > worst[1]:
            worst = (k, err)
    assert worst[1] <= 2e-4, f"gradient of {worst[0]} off by {worst[1]:.2e} (relative to its max + 1e-2)"
    gnorm = torch.nn.utils.clip_grad_norm_(model.parameters(), 1e9)
    assert abs(float(gnorm) - float(fx["gnorm"])) <= 2e-4 * float(fx["gnorm"])
    for k, b in model.named_buffers():
        if ".layer." in k or k.endswith("num_batches_tracked") or b.numel() == 0:
            continue
        assert np.abs(b.cpu().numpy() - fx["b::" + k]).max() <= 2e-5, k


@pytest.mark.parametrize("tag", ["T", "S"])
def test_train_step_through_the_bench_composition_matches_reference(tag):
    """the step `bench.py` times (SymbolWindows input + mural_amd.train.CrossEntropySum +
    mural_amd.train.clip_grad_norm_ over the flat gradient buffer, training.py:424-436) against the reference's own G7 fixture --
    loss, every gradient, the total gradient norm the clip returns, and the running statistics, at the fixture's 2e-4."""
    from mural_amd.data import SymbolWindows
    from mural_amd.train import CrossEntropySum, clip_grad_norm_
    fx = U.load(f"snv_train_{tag}.npz")
    model, _ = product_from_hp(fx["hp"])
    orc = U.snv_oracle_from_hp(fx["hp"], drops=(0.0, 0.0, 0.0))
    model.load_state_dict(U.snv_state_for(fx, orc))
    for m in model.modules():            # the fixture was generated with every dropout rate at 0
        if isinstance(m, nn.Dropout):
            m.p = 0.0
    model = model.cuda().train()
    cat = torch.from_numpy(fx["cat"]).cuda()
    codes = np.ascontiguousarray(fx["codes"]).astype(np.uint8)          # base codes 0..3 = MURAL_SYM_A..T: one symbol per column
    x = SymbolWindows(torch.from_numpy(codes).cuda())
    preds = model((torch.zeros(len(cat), 1, device="cuda"), cat), x)
    assert np.abs(preds.detach().cpu().numpy() - fx["preds"]).max() <= 2e-4
    loss = CrossEntropySum()(preds, torch.from_numpy(fx["y"]).cuda())
    model.zero_grad()
    loss.backward()
    assert abs(loss.item() - float(fx["loss"])) <= 1e-4 * abs(float(fx["loss"]))
    grads = {k: p.grad.clone() for k, p in model.named_parameters() if p.numel()}
    gnorm = clip_grad_norm_(model, 1e9)                                 # the flat-buffer route: must take it, not torch's fallback
    lay = model._train_layout
    assert lay.last_flat is not None and all(p.grad.data_ptr() == lay.last_flat.data_ptr() + 4 * o for p, o in zip(lay.plist, lay.poffs))
    assert abs(float(gnorm) - float(fx["gnorm"])) <= 2e-4 * float(fx["gnorm"])
    worst = ("", 0.0)
    for k, g in grads.items():
        if ".layer." in k:
            continue
        want = fx["g::" + k]
        scale = float(np.abs(want).max()) + 1e-2
        err = float(np.abs(g.cpu().numpy() - want).max()) / scale
        if err > worst[1]:
            worst = (k, err)
        assert torch.equal(g, dict(model.named_parameters())[k].grad)   # max_norm far above the norm: the clip leaves them alone
    assert worst[1] <= 2e-4, f"gradient of {worst[0]} off by {worst[1]:.2e} (relative to its max + 1e-2)"
    for k, b in model.named_buffers():
        if ".layer." in k or k.endswith("num_batches_tracked") or b.numel() == 0:
            continue
        assert np.abs(b.cpu().numpy() - fx["b::" + k]).max() <= 2e-5, k
    # and the clip proper: a max_norm below the norm scales every gradient by max_norm / (norm + 1e-6) (torch's rule, training.py:430)
    g0 = {k: p.grad.clone() for k, p in model.named_parameters() if p.numel()}
    total = clip_grad_norm_(model, float(gnorm) / 4)
    f = float(gnorm) / 4 / (float(total) + 1e-6)
    for k, p in model.named_parameters():
        if p.numel():
            assert torch.allclose(p.grad, g0[k] * f, rtol=1e-6, atol=0)


_OPS_ROUTE = {"MURAL_TRAIN_LOCAL_OPS": "1", "MURAL_TRAIN_HEAD_OPS": "1", "MURAL_TRAIN_FIRST_SEPARATE": "1", "MURAL_TRAIN_NO_FIRST_FOLD": "1",
              "MURAL_DEBUG_FIRST_SCATTER": "1", "MURAL_TRAIN_NO_POOL_FOLD": "1", "MURAL_TRAIN_NO_MID_FOLD": "1"}


@pytest.mark.parametrize("tag", ["T", "S"])
def test_train_step_per_op_route_matches_reference(tag, monkeypatch):
    """The forms the round-5 fusions replaced stay in the library behind switches (A/B runs): the local branch and the towers' heads as
    one launch per op, one histogram / table launch per tower, every stage-end and pool-side BatchNorm-backward apply as a pass of its
    own, the first layer's backward as an LDS-atomic scatter.  Both routes must meet the reference's G7 fixture."""
    for k, v in _OPS_ROUTE.items():
        monkeypatch.setenv(k, v)
    fx = U.load(f"snv_train_{tag}.npz")
    model, _ = product_from_hp(fx["hp"])
    orc = U.snv_oracle_from_hp(fx["hp"], drops=(0.0, 0.0, 0.0))
    model.load_state_dict(U.snv_state_for(fx, orc))
    for m in model.modules():
        if isinstance(m, nn.Dropout):
            m.p = 0.0
    model = model.cuda().train()
    cat = torch.from_numpy(fx["cat"]).cuda()
    x = U.onehot(fx["codes"]).cuda()
    preds = model((torch.zeros(len(cat), 1, dtype=torch.float64, device="cuda"), cat), x)
    assert np.abs(preds.detach().cpu().numpy() - fx["preds"]).max() <= 2e-4
    loss = nn.CrossEntropyLoss(reduction="sum")(preds, torch.from_numpy(fx["y"]).cuda())
    model.zero_grad()
    loss.backward()
    worst = ("", 0.0)
    for k, p in model.named_parameters():
        if ".layer." in k or p.numel() == 0:
            continue
        want = fx["g::" + k]
        err = float(np.abs(p.grad.cpu().numpy() - want).max()) / (float(np.abs(want).max()) + 1e-2)
        if err > worst[1]:
            worst = (k, err)
    assert worst[1] <= 2e-4, f"gradient of {worst[0]} off by {worst[1]:.2e} (relative to its max + 1e-2)"
    for k, b in model.named_buffers():
        if ".layer." in k or k.endswith("num_batches_tracked") or b.numel() == 0:
            continue
        assert np.abs(b.cpu().numpy() - fx["b::" + k]).max() <= 2e-5, k


def test_fused_and_per_op_routes_agree_with_dropout_on(monkeypatch):
    """Dropout inside the fused kernels (the local branch's three masks, the towers' distal_fc masks): the fused launches draw the masks
    of the per-op route -- same counter-based generator, same element index, same seeds -- so with every dropout at its default rate
    (0.1 / 0.1 / 0.25) both routes give the same outputs and the same gradients up to the order of their float sums, forward and
    backward masks included (a mask mismatch between the directions would move gradients by tens of per cent)."""
    fx = U.load("snv_train_S.npz")
    cat = torch.from_numpy(fx["cat"]).cuda()
    x = U.onehot(fx["codes"]).cuda()
    y = torch.from_numpy(fx["y"]).cuda()
    results = []
    for ops in (False, True):
        for k, v in _OPS_ROUTE.items():
            if ops:
                monkeypatch.setenv(k, v)
            else:
                monkeypatch.delenv(k, raising=False)
        model, _ = product_from_hp(fx["hp"])
        model.load_state_dict(U.snv_state_for(fx, U.snv_oracle_from_hp(fx["hp"])))
        model = model.cuda().train()
        assert any(m.p > 0 for m in model.modules() if isinstance(m, nn.Dropout))
        torch.manual_seed(11)
        out = model((torch.zeros(len(cat), 1, device="cuda"), cat), x)
        nn.CrossEntropyLoss(reduction="sum")(out, y).backward()
        results.append((out.detach().clone(), {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.numel()},
                        {k: b.detach().clone() for k, b in model.named_buffers() if b.is_floating_point() and b.numel()}))
    (o1, g1, b1), (o2, g2, b2) = results
    assert float((o1 - o2).abs().max()) <= 2e-5
    for k in g1:
        # (the absolute floor of the G7 tests: biases in front of a batch-statistics BatchNorm have a mathematically zero gradient)
        scale = float(g1[k].abs().max()) + 1e-2
        assert float((g1[k] - g2[k]).abs().max()) <= 2e-4 * scale, k
    for k in b1:
        assert float((b1[k] - b2[k]).abs().max()) <= 1e-5 * (float(b1[k].abs().max()) + 1e-3), k


@pytest.mark.parametrize("hp,B", [((10, 3, 1000, 150, 75, 32, 3, 4, 2), 37), ((5, 3, 100, 150, 75, 32, 3, 2, 2), 2), ((7, 3, 200, 40, 9, 32, 3, 8, 0), 53),
                                  ((4, 2, 128, 200, 130, 32, 3, 3, 2), 19), ((10, 3, 1000, 150, 75, 32, 3, 4, 1), 21)])
def test_fused_and_per_op_routes_agree_on_ragged_shapes(hp, B, monkeypatch):
    """The fused launches of round 5 (local branch, towers' heads, first-layer backward with its fold, histograms) against the per-op
    launches they replace, on shapes the fixtures do not have: batches that are no multiple of a 16-row tile (37, 53, 19, 21) and the
    smallest batch the step takes (2), two / three / eight classes, other hidden widths and k-mer orders (Linear inputs of 40, 45, 95
    features; reduction runs of 10, 12, 24, 33, 38, 50 steps), Network0 / 1 / 2; dropouts at their defaults, weights from the oracle's
    initialisation.  Outputs, every gradient and every running statistic must agree up to the order of the float sums."""
    hp = np.array(hp)
    rng = np.random.default_rng(int(hp.sum()) + B)
    r, order, R = int(hp[0]), int(hp[1]), int(hp[2])
    ncol = 2 * r + 1 - (order - 1)
    cat = torch.from_numpy(rng.integers(0, 4 ** order + 1, size=(B, ncol))).cuda()
    codes = rng.integers(0, 4, size=(B, 2 * R + 1)).astype(np.uint8)
    codes[rng.integers(0, B, 5), rng.integers(0, 2 * R + 1, 5)] = 4          # a few N
    x = U.onehot(codes).cuda()
    y = torch.from_numpy(rng.integers(0, int(hp[7]), size=B)).cuda()
    orc = U.snv_oracle_from_hp(hp)
    sd = {k: v.clone() for k, v in orc.state_dict().items()}
    results = []
    for ops in (False, True):
        for k, v in _OPS_ROUTE.items():
            if ops:
                monkeypatch.setenv(k, v)
            else:
                monkeypatch.delenv(k, raising=False)
        model, _ = product_from_hp(hp)
        model.load_state_dict(sd)
        model = model.cuda().train()
        torch.manual_seed(23)
        out = model((torch.zeros(B, 1, device="cuda"), cat), x)
        nn.CrossEntropyLoss(reduction="sum")(out, y).backward()
        results.append((out.detach().clone(), {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.numel() and p.grad is not None},
                        {k: b.detach().clone() for k, b in model.named_buffers() if b.is_floating_point() and b.numel()}))
    (o1, g1, b1), (o2, g2, b2) = results
    # (a batch of two rows: batch-statistics BatchNorm over two samples amplifies float32 round-off -- both routes sit 1.9e-2 from a
    # float64 evaluation of the same step there, tools/r5_b2check.py)
    gtol = 5e-4 if B >= 16 else 2e-2
    assert torch.isfinite(o1).all() and float((o1 - o2).abs().max()) <= (5e-5 if B >= 16 else 1e-3) * (float(o2.abs().max()) + 1.0)
    assert set(g1) == set(g2) and len(g1) >= 6
    for k in g1:
        scale = float(g2[k].abs().max()) + 1e-2
        assert float((g1[k] - g2[k]).abs().max()) <= gtol * scale, k
    for k in b1:
        assert float((b1[k] - b2[k]).abs().max()) <= 1e-5 * (float(b2[k].abs().max()) + 1e-3), k


def test_direct_gradient_mode_steps_aside_for_hooks_and_kept_gradients():
    """the step sets p.grad itself (model/train_step.py).  A parameter hook must still fire (the step then routes the
    gradients through autograd), gradients a caller keeps across zero_grad(set_to_none=True) must not be rewritten by the next
    backward, and a second backward over the same graph must fail with a clear message."""
    fx = U.load("snv_train_T.npz")
    model, _ = product_from_hp(fx["hp"])
    model.load_state_dict(U.snv_state_for(fx, U.snv_oracle_from_hp(fx["hp"])))
    model = model.cuda().train()
    cat = torch.from_numpy(fx["cat"]).cuda()
    x = U.onehot(fx["codes"]).cuda()
    y = torch.from_numpy(fx["y"]).cuda()
    crit = nn.CrossEntropyLoss(reduction="sum")
    cont = torch.zeros(len(cat), 1, device="cuda")
    torch.manual_seed(3)
    crit(model((cont, cat), x), y).backward()
    kept = {k: p.grad for k, p in model.named_parameters() if p.numel()}           # the caller keeps last step's gradients
    snap = {k: g.clone() for k, g in kept.items()}
    model.zero_grad(set_to_none=True)
    torch.manual_seed(4)
    out = model((cont, cat), x)
    loss = crit(out, y)
    loss.backward(retain_graph=True)
    assert all(torch.equal(kept[k], snap[k]) for k in kept), "the next backward rewrote gradients the caller still holds"
    assert any(not torch.equal(p.grad, snap[k]) for k, p in model.named_parameters() if p.numel())
    with pytest.raises(RuntimeError, match="ONE backward"):
        loss.backward()
    # a DERIVED tensor (a flattened view, a row, a detach()) shares the memory without being one of the view objects:
    # the storage's holder count sees it, and the next backward steps aside for it as well
    from mural_amd.model import train_step as TS
    for it, derive in enumerate((lambda g: g.view(-1), lambda g: g[0], lambda g: g.detach())):
        w = model.conv1_2[1].weight
        held = derive(w.grad)
        snap_held = held.clone()
        model.zero_grad(set_to_none=True)
        torch.manual_seed(5 + it)      # (another dropout mask: other gradients)
        crit(model((cont, cat), x), y).backward()
        assert torch.equal(held, snap_held), "the next backward rewrote memory a derived tensor still holds"
        assert not torch.equal(derive(w.grad), snap_held)
        del held
    # ... and with nothing held the buffer IS reused (the point of the direct mode)
    lay = TS._layout(model)
    model.zero_grad(set_to_none=True)
    crit(model((cont, cat), x), y).backward()
    ptr = lay.own_flat.data_ptr()
    model.zero_grad(set_to_none=True)
    crit(model((cont, cat), x), y).backward()
    assert lay.own_flat.data_ptr() == ptr
    # a tensor hook on one parameter: the whole step goes through autograd's accumulation and the hook sees its gradient
    model.zero_grad(set_to_none=True)
    seen = []
    w = model.conv1_2[1].weight
    h = w.register_hook(lambda g: seen.append(g.clone()))
    torch.manual_seed(4)
    crit(model((cont, cat), x), y).backward()
    h.remove()
    assert len(seen) == 1 and torch.equal(seen[0], w.grad)


def test_train_step_at_batch_256_matches_reference():
    """G7 at B = 256 (dropouts 0): the reference's own loss, gradients and running statistics for a batch between the flip-free
    32-row fixture (2e-4) and the float64-judged batch of 4096 (3e-2).  At 8 x the max-pool windows and ReLU inputs no weight seed
    keeps every decision of the forward clear of float32 round-off (the fixture records its margins, 2.8e-7 / 2.4e-7, and that the
    reference's own float32 gradients sit 1.9e-4 from a float64 evaluation): a window or a ReLU may resolve the other way than in
    the reference's summation order, and each such flip moves a gradient by a few 1e-3 of its size at most -- hence 5e-3."""
    fx = U.load("snv_train_S256.npz")
    model, _ = product_from_hp(fx["hp"])
    orc = U.snv_oracle_from_hp(fx["hp"], drops=(0.0, 0.0, 0.0))
    model.load_state_dict(U.snv_state_for(fx, orc))
    for m in model.modules():
        if isinstance(m, nn.Dropout):
            m.p = 0.0
    model = model.cuda().train()
    cat = torch.from_numpy(fx["cat"]).cuda()
    x = U.onehot(fx["codes"]).cuda()
    assert len(cat) == 256
    preds = model((torch.zeros(len(cat), 1, dtype=torch.float64, device="cuda"), cat), x)
    assert np.abs(preds.detach().cpu().numpy() - fx["preds"]).max() <= 2e-4
    loss = nn.CrossEntropyLoss(reduction="sum")(preds, torch.from_numpy(fx["y"]).cuda())
    model.zero_grad()
    loss.backward()
    assert abs(loss.item() - float(fx["loss"])) <= 1e-4 * abs(float(fx["loss"]))
    worst = ("", 0.0)
    for k, p in model.named_parameters():
        if ".layer." in k or p.numel() == 0:
            continue
        want = fx["g::" + k]
        scale = float(np.abs(want).max()) + 1e-2
        err = float(np.abs(p.grad.cpu().numpy() - want).max()) / scale
        if err > worst[1]:
            worst = (k, err)
    assert worst[1] <= 5e-3, f"gradient of {worst[0]} off by {worst[1]:.2e} (relative to its max + 1e-2)"
    for k, b in model.named_buffers():
        if ".layer." in k or k.endswith("num_batches_tracked") or b.numel() == 0:
            continue
        assert np.abs(b.cpu().numpy() - fx["b::" + k]).max() <= 2e-5, k


def test_dropout_statistics_and_determinism():
    from mural_amd.model import train_ops as T
    x = torch.ones(1 << 20, device="cuda")
    y = T.Dropout.apply(x, 0.25, 1234)
    keep = float((y != 0).float().mean())
    assert abs(keep - 0.75) < 5e-3
    assert torch.allclose(y[y != 0], torch.full((1,), 1 / 0.75, device="cuda"))
    assert torch.equal(y, T.Dropout.apply(x, 0.25, 1234)) and not torch.equal(y, T.Dropout.apply(x, 0.25, 1235))
    # a device-resident counter is added to the seed (graph replay draws new masks from it)
    ctr = torch.tensor([1], dtype=torch.int64, device="cuda")
    assert torch.equal(T.Dropout.apply(x, 0.25, 1234, ctr), T.Dropout.apply(x, 0.25, 1235))


def test_optimizer_step_runs_and_eval_uses_updated_weights():
    """one Adam step through the drop-in module, then the fused eval kernels see the new parameters/statistics"""
    fx = U.load("snv_train_T.npz")
    model, _ = product_from_hp(fx["hp"])
    orc = U.snv_oracle_from_hp(fx["hp"])
    sd = U.snv_state_for(fx, orc)
    model.load_state_dict(sd)
    model = model.cuda().train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    cat = torch.from_numpy(fx["cat"]).cuda()
    x = U.onehot(fx["codes"]).cuda()
    y = torch.from_numpy(fx["y"]).cuda()
    crit = nn.CrossEntropyLoss(reduction="sum")
    torch.manual_seed(0)
    losses = []
    for _ in range(3):
        preds = model((torch.zeros(len(cat), 1, device="cuda"), cat), x)
        loss = crit(preds, y)
        opt.zero_grad()
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 10)
        opt.step()
        losses.append(loss.item())
    assert all(np.isfinite(losses))
    model.eval()
    orc.load_state_dict({k: v.cpu() for k, v in model.state_dict().items()})
    orc.eval()
    with torch.no_grad():
        got = model((torch.zeros(len(cat), 1, device="cuda"), cat), x).cpu().numpy()
        want = orc((torch.zeros(len(cat), 1, dtype=torch.float64), cat.cpu()), x.cpu()).numpy()
    assert np.abs(np.exp(got) - np.exp(want)).max() <= 1e-5


@pytest.mark.parametrize("model_no", [0, 1])
def test_train_step_network0_network1_vs_oracle_autograd(model_no):
    """sub-graphs of Network2: gradients against the CPU oracle's autograd (the oracle is pinned for training by G7)"""
    hp = np.array([5, 3, 100, 150, 75, 32, 3, 4, model_no])
    model, _ = product_from_hp(hp)
    orc = U.snv_oracle_from_hp(hp, drops=(0.0, 0.0, 0.0))
    from oracle import synth
    sd = synth.synth_state_dict(orc.state_dict(), 99)
    orc.load_state_dict(sd)
    model.load_state_dict(sd)
    for m in model.modules():
        if isinstance(m, nn.Dropout):
            m.p = 0.0
    model = model.cuda().train()
    orc.train()
    rng = np.random.default_rng(5)
    B = 24
    codes = rng.integers(0, 4, size=(B, 201)).astype(np.uint8)
    cat = torch.from_numpy(rng.integers(0, 65, size=(B, 9)).astype(np.int64))
    y = torch.from_numpy(rng.integers(0, 4, size=B))
    x = U.onehot(codes)
    crit = nn.CrossEntropyLoss(reduction="sum")
    want = crit(orc((torch.zeros(B, 1, dtype=torch.float64), cat), x), y)
    want.backward()
    got = crit(model((torch.zeros(B, 1, device="cuda"), cat.cuda()), x.cuda()), y.cuda())
    got.backward()
    assert abs(got.item() - want.item()) <= 1e-4 * abs(want.item())
    ref = dict(orc.named_parameters())
    for k, p in model.named_parameters():
        if ".layer." in k or p.numel() == 0:
            continue
        w = ref[k].grad.numpy()
        assert np.abs(p.grad.cpu().numpy() - w).max() <= 2e-4 * (np.abs(w).max() + 1e-2), k


def test_train_step_with_local_order_6_embedding_table():
    """the reference CLI accepts any --local_order; at order 6 the embedding has 4 ** 6 + 1 = 4097 rows, whose gradient
    table (82 KB) is beyond the LDS the fused local-branch backward gets -- the step takes the per-op launches for such shapes
    instead of failing the launch.  Network0 and Network2 against the oracle's autograd."""
    for model_no in (0, 2):
        hp = np.array([7, 6, 100, 150, 75, 32, 3, 4, model_no])
        model, _ = product_from_hp(hp)
        orc = U.snv_oracle_from_hp(hp, drops=(0.0, 0.0, 0.0))
        from oracle import synth
        sd = synth.synth_state_dict(orc.state_dict(), 606)
        orc.load_state_dict(sd)
        model.load_state_dict(sd)
        assert [v.shape[0] for k, v in model.state_dict().items() if k.endswith("emb_layer.weight")] == [4097]
        for m in model.modules():
            if isinstance(m, nn.Dropout):
                m.p = 0.0
        model = model.cuda().train()
        orc.train()
        rng = np.random.default_rng(6)
        B = 40
        codes = rng.integers(0, 4, size=(B, 201)).astype(np.uint8)
        cat = torch.from_numpy(rng.integers(0, 4097, size=(B, 10)).astype(np.int64))
        cat[0, :3] = 4096                                  # the padding row is trained as well (no padding_idx, model_snv.py:322)
        y = torch.from_numpy(rng.integers(0, 4, size=B))
        x = U.onehot(codes)
        crit = nn.CrossEntropyLoss(reduction="sum")
        want = crit(orc((torch.zeros(B, 1, dtype=torch.float64), cat), x), y)
        want.backward()
        got = crit(model((torch.zeros(B, 1, device="cuda"), cat.cuda()), x.cuda()), y.cuda())
        got.backward()
        assert abs(got.item() - want.item()) <= 1e-4 * abs(want.item())
        ref = dict(orc.named_parameters())
        for k, p in model.named_parameters():
            if ".layer." in k or p.numel() == 0:
                continue
            w = ref[k].grad.numpy()
            assert np.abs(p.grad.cpu().numpy() - w).max() <= 2e-4 * (np.abs(w).max() + 1e-2), (model_no, k)


def test_graphed_train_step_matches_eager():
    """hipGraph replay of the whole step (mural_amd.train.GraphedTrainStep) == the eager step: same parameters after the
    same batches (dropout off so both paths are deterministic), and a non-encoding input is reported one step late."""
    from mural_amd.train import GraphedTrainStep
    fx = U.load("snv_train_T.npz")
    B = len(fx["cat"])
    cat = torch.from_numpy(fx["cat"]).cuda()
    x = U.onehot(fx["codes"]).cuda()
    y = torch.from_numpy(fx["y"]).cuda()
    cont = torch.zeros(B, 1, dtype=torch.float64, device="cuda")
    crit = nn.CrossEntropyLoss(reduction="sum")

    def make():
        model, _ = product_from_hp(fx["hp"])
        orc = U.snv_oracle_from_hp(fx["hp"], drops=(0.0, 0.0, 0.0))
        model.load_state_dict(U.snv_state_for(fx, orc))
        for m in model.modules():
            if isinstance(m, nn.Dropout):
                m.p = 0.0
        model = model.cuda().train()
        # plain SGD: Adam divides by the gradient's own magnitude, so the 1e-5 run-to-run noise of the float atomics in small
        # gradients becomes full-size parameter steps and, through a max-pool near-tie, an occasional 1e-3 difference between two
        # EAGER runs -- not what this test is about (Adam under capture: test_eval_after_graph_replays_uses_current_weights)
        return model, torch.optim.SGD(model.parameters(), lr=2e-3, momentum=0.9)

    eager, opt_e = make()
    n_steps = 3 + 2            # GraphedTrainStep: 3 eager warm-up steps + 2 replays (capturing executes nothing), same batch
    for _ in range(n_steps):
        loss = crit(eager((cont, cat), x), y)
        opt_e.zero_grad()
        loss.backward()
        torch.nn.utils.clip_grad_norm_(eager.parameters(), 10)
        opt_e.step()
    graphed, opt_g = make()
    step = GraphedTrainStep(graphed, opt_g, crit, cont, cat, x, y)
    for _ in range(2):
        loss_g = step(cont, cat, x, y)
    step.finish()
    assert abs(loss_g.item() - loss.item()) <= 1e-3 * abs(loss.item())
    for (k, p), (_, q) in zip(eager.named_parameters(), graphed.named_parameters()):
        if not p.numel() or ("conv" in k and k.endswith(".bias")):
            continue   # conv biases in front of a batch-statistics BN have ~1e-7 noise gradients: Adam turns them into +-lr steps
        assert float((p.detach() - q.detach()).abs().max()) <= 1e-4 * (float(p.detach().abs().max()) + 1e-3), k
    bad = x.clone()
    bad[0, :, 5] = 0.3
    step(cont, cat, bad, y)
    with pytest.raises(ValueError, match="not a MuRaL"):
        step.finish()


def test_eval_after_graph_replays_uses_current_weights():
    """Graph replays update weights and BatchNorm running statistics without touching tensor versions: every train -> eval
    transition must rebuild the folded eval-mode copy.  replay, eval, replay, eval -- each eval against the oracle loaded with
    the model's state at that moment (and the two evals must differ)."""
    from mural_amd.train import GraphedTrainStep
    from tests.test_gpu_snv import assert_probs_close, product_from_hp
    fx = U.load("snv_train_T.npz")
    crit = nn.CrossEntropyLoss(reduction="sum")
    cat = torch.from_numpy(fx["cat"]).cuda()
    x = U.onehot(fx["codes"]).cuda()
    y = torch.from_numpy(fx["y"].astype(np.int64)).cuda()
    cont = torch.zeros(len(cat), 1, dtype=torch.float64, device="cuda")
    model, _ = product_from_hp(fx["hp"])
    orc = U.snv_oracle_from_hp(fx["hp"], drops=(0.0, 0.0, 0.0))
    model.load_state_dict(U.snv_state_for(fx, orc))
    for m in model.modules():
        if isinstance(m, nn.Dropout):
            m.p = 0.0
    model = model.cuda().train()
    opt = torch.optim.Adam(model.parameters(), lr=5e-3, capturable=True)
    step = GraphedTrainStep(model, opt, crit, cont, cat, x, y)
    outs = []
    for _ in range(2):
        model.train()
        for _ in range(3):
            step(cont, cat, x, y)
        step.finish()
        model.eval()
        with torch.no_grad():
            got = model((cont, cat), x).cpu().numpy()
        orc.load_state_dict({k: v.cpu() for k, v in model.state_dict().items()})
        orc.eval()
        with torch.no_grad():
            want = orc((cont.cpu(), cat.cpu()), x.cpu()).numpy()
        assert_probs_close(got, want, 2, "eval after replay")
        outs.append(got)
    assert np.abs(np.exp(outs[0]) - np.exp(outs[1])).max() > 1e-4


def test_conv32_kernels_match_torch_fp64():
    """The MFMA conv kernels behind the training step (forward with pre-op / residuals / fused batch sums, input gradient
    with BatchNorm-backward sums, weight gradient, fused backward) against torch ops in float64 on the CPU, over full,
    partial and single-row tiles."""
    import subprocess
    import sys
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "gpu_debug_conv32.py")
    out = subprocess.run([sys.executable, tool], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "worst" in out.stdout


def test_channel_last_conv_kernels_match_torch_fp64():
    """The channel-last [B][L][32] conv kernels the composed step runs on (csrc/conv32_cl.hip): forward with BatchNorm finalisation,
    residuals and fused batch sums; backward with weight / bias gradient, input gradient and BatchNorm-backward sums, with and
    without the ReLU in front, over one-tile, many-tile and ragged-tail batches."""
    import subprocess
    import sys
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "gpu_debug_conv32_cl.py")
    out = subprocess.run([sys.executable, tool], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "worst" in out.stdout


def test_wave_private_conv_kernels_match_torch_fp64():
    """The wave-private conv kernels of the composed step (csrc/conv32_wave.hip) through the same harness: forward with BatchNorm
    finalisation, residuals and fused batch sums; backward whose BatchNorm-backward sums and weight gradient come out of the
    partial-row algebra (dW from sums of dy (x) act(x) and three vectors of dy sums) -- unit geometries of 4 .. 9 blocks, several
    rows per unit, ragged last units, rows of one column, the widest row a wave's image holds (142)."""
    import subprocess
    import sys
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "gpu_debug_conv32_cl.py")
    out = subprocess.run([sys.executable, tool], capture_output=True, text=True, timeout=600, env=dict(os.environ, WHICH="cw"))
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "worst" in out.stdout and "L= 142" in out.stdout


def test_train_step_conv_gradients_are_bitwise_reproducible():
    """Two runs of the same step (S fixture tiled to 1023 rows: many workgroups, a ragged last unit) give bit-identical losses and
    bit-identical gradients of every 32 -> 32 conv layer, its BatchNorm and the tower heads: units are assigned to waves by index,
    partial rows are summed in a fixed order, batch sums meet in float64.  (The first layer's gradient table and the embedding
    gradient are float atomics in LDS: last-bit differences between runs are theirs.)"""
    fx = U.load("snv_train_S.npz")
    reps = 86
    cat = torch.from_numpy(np.tile(fx["cat"], (reps, 1))[:1023]).cuda()
    x = U.onehot(np.tile(fx["codes"], (reps, 1))[:1023]).cuda()
    y = torch.from_numpy(np.tile(fx["y"], reps)[:1023]).cuda()
    runs = []
    for _ in range(2):
        model, _m = product_from_hp(fx["hp"])
        orc = U.snv_oracle_from_hp(fx["hp"], drops=(0.0, 0.0, 0.0))
        model.load_state_dict(U.snv_state_for(fx, orc))
        for m in model.modules():
            if isinstance(m, nn.Dropout):
                m.p = 0.0
        model = model.cuda().train()
        preds = model((torch.zeros(len(cat), 1, dtype=torch.float64, device="cuda"), cat), x)
        loss = nn.CrossEntropyLoss(reduction="sum")(preds, y)
        loss.backward()
        runs.append((loss.item(), {k: p.grad.cpu().numpy().copy() for k, p in model.named_parameters() if p.grad is not None}))
    assert runs[0][0] == runs[1][0]
    differ = [k for k, g in runs[0][1].items() if not np.array_equal(g, runs[1][1][k])]
    fixed = [k for k in differ if k.startswith(("RBs", "conv2", "conv3", "distal_fc"))]
    assert not fixed, (fixed, differ)


@pytest.mark.parametrize("name", ["snv_synth_generic_c16k5_net2.npz", "snv_synth_generic_c24k4_net2.npz", "snv_synth_generic_c64k3_net1.npz"])
def test_train_step_other_channel_and_kernel_sizes(name):
    """CNN_out_channels / CNN_kernel_size other than 32 / 3 train on the general per-layer ops: one step (dropouts 0) against
    the oracle's autograd on the same weights and inputs (the oracle reproduces the reference's forward for these shapes,
    tests/test_oracle_golden.py)."""
    from tests.test_gpu_snv import product_from_hp
    fx = U.load(name)
    model, model_no = product_from_hp(fx["hp"])
    orc = U.snv_oracle_from_hp(fx["hp"], drops=(0.0, 0.0, 0.0))
    sd = U.snv_state_for(fx, orc)
    model.load_state_dict(sd)
    orc.load_state_dict(sd)
    model = model.cuda().train()
    orc.train()
    for m in model.modules():
        if isinstance(m, nn.Dropout):
            m.p = 0.0
    cat, x = torch.from_numpy(fx["cat"]), U.onehot(fx["codes"])
    y = torch.from_numpy(np.arange(len(cat)) % int(fx["hp"][7]))
    crit = nn.CrossEntropyLoss(reduction="sum")
    want = orc((torch.zeros(len(cat), 1, dtype=torch.float64), cat), x)
    crit(want, y).backward()
    got = model((torch.zeros(len(cat), 1, dtype=torch.float64).cuda(), cat.cuda()), x.cuda())
    loss = crit(got, y.cuda())
    loss.backward()
    assert np.abs(got.detach().cpu().numpy() - want.detach().numpy()).max() <= 2e-4
    ref_grads = dict(orc.named_parameters())
    for k, p in model.named_parameters():
        if ".layer." in k or p.numel() == 0 or ref_grads[k].grad is None:
            continue
        w = ref_grads[k].grad.numpy()
        err = float(np.abs(p.grad.cpu().numpy() - w).max()) / (float(np.abs(w).max()) + 1e-2)
        assert err <= 2e-4, (k, err)
    ref_buf = dict(orc.named_buffers())
    for k, b in model.named_buffers():
        if ".layer." in k or k.endswith("num_batches_tracked") or b.numel() == 0:
            continue
        assert np.abs(b.cpu().numpy() - ref_buf[k].numpy()).max() <= 2e-5, k


def test_train_step_at_benchmark_batch_4096_vs_oracle_autograd():
    """BASELINE config 3's shape: one step of the S-config at batch 4096 (dropouts 0) against the oracle's autograd -- the
    multi-workgroup partial reductions, the 32-slot float64 BatchNorm accumulators and the first-layer gradient-table reduction
    all run at the size the benchmark uses."""
    from tests.test_gpu_snv import product_from_hp
    hp = np.array([10, 3, 1000, 150, 75, 32, 3, 4, 2])
    B = 4096
    rng = np.random.default_rng(4096)
    model, _ = product_from_hp(hp)
    orc = U.snv_oracle_from_hp(hp, drops=(0.0, 0.0, 0.0))
    from oracle import synth as _synth
    sd = _synth.synth_state_dict(orc.state_dict(), 4096)
    model.load_state_dict(sd)
    orc.load_state_dict(sd)
    for m in model.modules():
        if isinstance(m, nn.Dropout):
            m.p = 0.0
    model = model.cuda().train()
    orc.train()
    codes = rng.integers(0, 4, size=(B, 2001)).astype(np.uint8)
    codes[rng.integers(0, B, 50), rng.integers(0, 2001, 50)] = 4
    centre = codes[:, 990:1011].astype(np.int64)
    cat = np.where((centre[:, :-2] > 3) | (centre[:, 1:-1] > 3) | (centre[:, 2:] > 3), 64,
                   np.minimum(centre[:, :-2], 3) * 16 + np.minimum(centre[:, 1:-1], 3) * 4 + np.minimum(centre[:, 2:], 3))
    y = torch.from_numpy(rng.choice(4, size=B, p=[0.955, 0.015, 0.015, 0.015]))
    crit = nn.CrossEntropyLoss(reduction="sum")
    x = U.onehot(codes)
    cat_t = torch.from_numpy(cat)
    torch.set_num_threads(min(64, torch.get_num_threads()))
    # 4096 x 134 x 32 activations per layer: a handful of them sit within float32 round-off of a ReLU kink or a max-pool tie, and
    # which side they fall on depends on the summation order of the convolutions in front.  Each such flip moves a gradient tensor
    # by ~1e-3 of its maximum, so the reference's own float32 gradients sit 1e-4 .. 4e-3 from a float64 evaluation of the same
    # model, and so do ours (per layer the MFMA accumulation order measures 1.45 x the round-off of torch's CPU convolution,
    # tools/gpu_debug_conv32_cl.py).  "Equal to the reference" is therefore judged against float64: every HIP gradient within
    # 3e-2 of it (a wrong or dropped partial reduction is off by O(0.1 .. 1)) and, averaged over the tensors, within 4 x the distance
    # of torch's float32 path (WHICH elements flip is a coin toss of the summation order: with the conv launches' partition of the
    # batch of round 5 this seed measures 2.1 x, with round 6's -- fewer, longer partial sums -- 3.2 x; the flip-free fixtures
    # G7 T / S / S256 hold both at their 2e-4 / 5e-3).  The flip-free check at this size is
    # test_train_step_replicated_batch_equals_scaled_fixture below.
    import copy
    orc64 = copy.deepcopy(orc).double()
    want = orc((torch.zeros(B, 1, dtype=torch.float64), cat_t), x)
    loss_ref = crit(want, y)
    loss_ref.backward()
    crit(orc64((torch.zeros(B, 1, dtype=torch.float64), cat_t), x.double()), y).backward()
    got = model((torch.zeros(B, 1, dtype=torch.float64).cuda(), cat_t.cuda()), x.cuda())
    loss = crit(got, y.cuda())
    loss.backward()
    assert abs(loss.item() - loss_ref.item()) <= 2e-5 * abs(loss_ref.item())
    assert np.abs(np.exp(got.detach().cpu().numpy()) - np.exp(want.detach().numpy())).max() <= 2e-5
    ref_grads, exact = dict(orc.named_parameters()), dict(orc64.named_parameters())
    worst = ("", 0.0)
    errs_hip, errs_ref = [], []
    for k, p in model.named_parameters():
        if ".layer." in k or p.numel() == 0:
            continue
        w = ref_grads[k].grad.numpy().astype(np.float64)
        t = exact[k].grad.numpy()
        g = p.grad.cpu().numpy().astype(np.float64)
        scale = float(np.abs(t).max()) + 1e-2
        err_hip, err_ref = float(np.abs(g - t).max()) / scale, float(np.abs(w - t).max()) / scale
        if os.environ.get("MURAL_TEST_VERBOSE"):
            print(f"{k:32s} |exact| {np.abs(t).max():.3e} hip-vs-exact {err_hip:.2e} torch32-vs-exact {err_ref:.2e}")
        errs_hip.append(err_hip)
        errs_ref.append(err_ref)
        if err_hip > worst[1]:
            worst = (k, err_hip, err_ref)
    assert worst[1] <= 3e-2, f"gradient of {worst[0]}: {worst[1]:.2e} from the float64 value (the reference's float32 path: {worst[2]:.2e})"
    if os.environ.get("MURAL_TEST_VERBOSE"):
        print("mean distance from float64: hip %.3e torch32 %.3e" % (np.mean(errs_hip), np.mean(errs_ref)))
    assert np.mean(errs_hip) <= 4.0 * np.mean(errs_ref) + 1e-5, (np.mean(errs_hip), np.mean(errs_ref))
    ref_buf = dict(orc.named_buffers())
    for k, b in model.named_buffers():
        if ".layer." in k or k.endswith("num_batches_tracked") or b.numel() == 0:
            continue
        assert np.abs(b.cpu().numpy() - ref_buf[k].numpy()).max() <= 2e-5, k


def _replicated_batch_deviation(reps=128):
    """(worst tensor, its deviation / tolerance, loss deviation / tolerance, preds deviation / tolerance) of one step on `reps` copies of
    the S fixture's 32 rows against reps x the fixture's gradients (tolerance = the fixture's own 2e-4 bar)."""
    fx = U.load("snv_train_S.npz")
    model, _ = product_from_hp(fx["hp"])
    orc = U.snv_oracle_from_hp(fx["hp"], drops=(0.0, 0.0, 0.0))
    model.load_state_dict(U.snv_state_for(fx, orc))
    for m in model.modules():
        if isinstance(m, nn.Dropout):
            m.p = 0.0
    model = model.cuda().train()
    cat = torch.from_numpy(np.tile(fx["cat"], (reps, 1))).cuda()
    x = U.onehot(np.tile(fx["codes"], (reps, 1))).cuda()
    y = torch.from_numpy(np.tile(fx["y"], reps)).cuda()
    preds = model((torch.zeros(len(cat), 1, dtype=torch.float64, device="cuda"), cat), x)
    d_preds = float(np.abs(preds.detach().cpu().numpy() - np.tile(fx["preds"], (reps, 1))).max()) / 2e-4
    loss = nn.CrossEntropyLoss(reduction="sum")(preds, y)
    model.zero_grad()
    loss.backward()
    d_loss = abs(loss.item() - reps * float(fx["loss"])) / (1e-4 * reps * abs(float(fx["loss"])))
    worst = ("", 0.0)
    for k, p in model.named_parameters():
        if ".layer." in k or p.numel() == 0:
            continue
        want = fx["g::" + k] * reps
        tol = 2e-4 * (float(np.abs(want).max()) + 1e-2 * reps)
        err = float(np.abs(p.grad.cpu().numpy() - want).max()) / tol
        if err > worst[1]:
            worst = (k, err)
    return worst[0], worst[1], d_loss, d_preds


def test_train_step_replicated_batch_equals_scaled_fixture():
    """Size-independent property at the benchmark's batch size: 128 copies of the 32 rows of the S fixture (B = 4096) have the batch
    statistics of the 32 rows, so the outputs repeat and the CE-sum loss and every gradient are 128 x the fixture's -- to the
    fixture's own tolerance, because the fixture is clear of max-pool near-ties (pool_margin) and every copy of a row takes the same
    side of every ReLU.  Runs the many-workgroup partial-row reductions, the 32-slot float64 BatchNorm accumulators, the ragged last
    tiles and the first-layer gradient table at full size without the float32 chaos of random rows."""
    name, dev, d_loss, d_preds = _replicated_batch_deviation()
    assert d_preds <= 1.0 and d_loss <= 1.0
    assert dev <= 1.0, f"gradient of {name}: {dev:.2f} x the allowed deviation"


@pytest.mark.parametrize("job", [0, 7, 13])
def test_a_dropped_partial_row_is_caught_at_the_benchmark_batch(job, monkeypatch):
    """VERDICT r02 (weak 2): the float64 comparison at B = 4096 allows 3e-2 per gradient, and ONE partial row of a conv layer's
    weight-gradient sum is ~1/30 of that layer's rows.  Fault injection (MURAL_DEBUG_DROP_PART_ROW leaves the last partial row of
    one layer out of its reduction): the replicated-batch property must flag it, far outside its tolerance -- that test, not the 3e-2
    bound, is what guards the partial reductions at this size."""
    clean = _replicated_batch_deviation()
    assert clean[1] <= 1.0
    monkeypatch.setenv("MURAL_DEBUG_DROP_PART_ROW", str(job))
    name, dev, _, _ = _replicated_batch_deviation()
    monkeypatch.delenv("MURAL_DEBUG_DROP_PART_ROW")
    assert dev > 5.0, f"a dropped partial row of job {job} moved {name} by only {dev:.2f} x the tolerance: the test would miss it"
    assert "conv" in name or "RBs" in name, name


def test_flat_clip_grad_norm_matches_torch():
    """mural_amd.train.clip_grad_norm_ (one reduction over the flat gradient buffer of the one-call backward) against
    torch.nn.utils.clip_grad_norm_ on the same gradients: same total norm, same clipped gradients; and the fall-back when a
    gradient was replaced."""
    from mural_amd.train import clip_grad_norm_
    fx = U.load("snv_train_T.npz")
    model, _ = product_from_hp(fx["hp"])
    orc = U.snv_oracle_from_hp(fx["hp"], drops=(0.0, 0.0, 0.0))
    model.load_state_dict(U.snv_state_for(fx, orc))
    model = model.cuda().train()
    cat = torch.from_numpy(fx["cat"]).cuda()
    x = U.onehot(fx["codes"]).cuda()
    y = torch.from_numpy(fx["y"]).cuda()
    crit = nn.CrossEntropyLoss(reduction="sum")

    def grads():
        model.zero_grad()
        crit(model((torch.zeros(len(cat), 1, dtype=torch.float64, device="cuda"), cat), x), y).backward()

    for max_norm in (1e9, 0.5):
        torch.manual_seed(5)
        grads()
        want_total = torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm)
        want = {k: p.grad.clone() for k, p in model.named_parameters() if p.numel()}
        torch.manual_seed(5)
        grads()
        assert model._train_layout.last_flat is not None
        total = clip_grad_norm_(model, max_norm)
        assert abs(float(total) - float(want_total)) <= 1e-5 * float(want_total)
        for k, p in model.named_parameters():
            if p.numel():
                # two separate backward passes: the float atomics of the first-layer gradient table land in a different order
                assert float((p.grad - want[k]).abs().max()) <= 2e-5 * (float(want[k].abs().max()) + 1e-6), k
    # a diverged step stays visible: an infinite gradient gives an infinite norm, coefficient 0 -- the finite gradients
    # become 0 and the infinite one NaN, as in torch; a NaN gradient gives a NaN norm and NaN everywhere
    for poison in (float("inf"), float("nan")):
        torch.manual_seed(5)
        grads()
        victim = next(p for p in model.parameters() if p.numel() > 8)
        victim.grad.view(-1)[3] = poison
        want_list = [p.grad.clone() for p in model.parameters() if p.numel()]
        total_ref = torch.linalg.vector_norm(torch.stack([g.norm() for g in want_list]))
        coef = torch.clamp(10.0 / (total_ref + 1e-6), max=1.0)
        want_list = [g * coef for g in want_list]
        total = clip_grad_norm_(model, 10.0)
        assert model._train_layout.last_flat is not None
        assert (torch.isnan(total) and torch.isnan(total_ref)) or float(total) == float(total_ref)
        for p, w in zip([p for p in model.parameters() if p.numel()], want_list):
            assert torch.equal(torch.isnan(p.grad), torch.isnan(w))
            assert torch.equal(torch.nan_to_num(p.grad), torch.nan_to_num(w))
    # a gradient that left the flat buffer: torch's path serves the call
    torch.manual_seed(5)
    grads()
    first = next(p for p in model.parameters() if p.numel())
    first.grad = first.grad.clone() * 2
    ref_total = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in model.parameters() if p.numel()))
    total = clip_grad_norm_(model, 1e9)
    assert abs(float(total) - float(ref_total)) <= 1e-5 * float(ref_total)


def test_train_step_does_not_read_unwritten_workspace(monkeypatch):
    """The composed training step keeps its activations, arg-max bytes, gradient tables and partial sums in ONE workspace from
    torch.empty: the same step with that workspace (and the gradient buffer's padding) pre-filled with 0xFF bytes gives the same
    loss and gradients."""
    from mural_amd.model import train_step as TS
    from tests.test_gpu_snv import _PoisonedTorch
    fx = U.load("snv_train_T.npz")
    model, _ = product_from_hp(fx["hp"])
    orc = U.snv_oracle_from_hp(fx["hp"], drops=(0.0, 0.0, 0.0))
    model.load_state_dict(U.snv_state_for(fx, orc))
    for m in model.modules():
        if isinstance(m, nn.Dropout):
            m.p = 0.0
    model = model.cuda().train()
    cat = torch.from_numpy(fx["cat"]).cuda()
    x = U.onehot(fx["codes"]).cuda()
    y = torch.from_numpy(fx["y"].astype(np.int64)).cuda()
    cont = torch.zeros(len(cat), 1, dtype=torch.float64, device="cuda")
    crit = nn.CrossEntropyLoss(reduction="sum")

    def step():
        loss = crit(model((cont, cat), x), y)
        model.zero_grad()
        loss.backward()
        return loss.item(), {k: p.grad.clone() for k, p in model.named_parameters() if p.numel()}

    l0, g0 = step()
    monkeypatch.setattr(TS, "torch", _PoisonedTorch())
    l1, g1 = step()
    assert abs(l1 - l0) <= 1e-5 * abs(l0)
    for k in g0:
        assert torch.isfinite(g1[k]).all(), k
        assert float((g1[k] - g0[k]).abs().max()) <= 2e-4 * (float(g0[k].abs().max()) + 1e-2), k


def test_train_step_writes_stay_inside_their_workspace_regions(monkeypatch):
    """The composed training step's workspace with 4 KB of poisoned guard bytes behind every region (activations, arg-max bytes,
    BatchNorm states, gradient tables, partial sums; the accumulator blocks count as one region): every guard byte survives a
    forward + backward."""
    import ctypes as C
    from mural_amd import _lib
    from mural_amd.model import train_step as TS
    from tests.test_gpu_snv import _PoisonedTorch
    guard = 4096
    monkeypatch.setenv("MURAL_DEBUG_WS_GUARD", str(guard))
    fx = U.load("snv_train_T.npz")
    model, _ = product_from_hp(fx["hp"])
    orc = U.snv_oracle_from_hp(fx["hp"], drops=(0.0, 0.0, 0.0))
    model.load_state_dict(U.snv_state_for(fx, orc))
    model = model.cuda().train()
    cat = torch.from_numpy(fx["cat"]).cuda()
    x = U.onehot(fx["codes"]).cuda()
    y = torch.from_numpy(fx["y"].astype(np.int64)).cuda()
    cont = torch.zeros(len(cat), 1, dtype=torch.float64, device="cuda")
    kept = []

    class Keep(_PoisonedTorch):
        @staticmethod
        def empty(*a, **k):
            t = _PoisonedTorch.empty(*a, **k)
            if t.dtype is torch.uint8:
                kept.append(t)
            return t

    monkeypatch.setattr(TS, "torch", Keep())
    loss = nn.CrossEntropyLoss(reduction="sum")(model((cont, cat), x), y)
    model.zero_grad()
    loss.backward()
    torch.cuda.synchronize()
    assert torch.isfinite(loss) and len(kept) == 1
    layout = (C.c_size_t * 512)()
    n_regions = _lib.lib().mural_debug_last_ws_layout(layout, 256)
    assert n_regions > 60
    ws = kept[0].cpu().numpy()
    acc_like = 0
    for i in range(n_regions):
        off, size = layout[2 * i], layout[2 * i + 1]
        nxt = layout[2 * i + 2] if i + 1 < n_regions else len(ws)
        if nxt - (off + size) < guard:          # inside the accumulator range: no guard between its blocks
            acc_like += 1
            continue
        zone = ws[off + size:off + size + guard]
        assert (zone == 255).all(), f"region {i}: a kernel wrote behind its {size} bytes"
    assert acc_like < n_regions // 2


def test_symbol_windows_route_equals_the_dense_route_bit_for_bit():
    """PackedGenome.encode_symbols -> model(local, SymbolWindows) in training mode: the training step's first layer works from symbols,
    the dense route only recovers them from the one-hot tensor -- outputs and BatchNorm statistics of the two routes are identical bit
    for bit, the gradients up to the order of their atomic sums (both strands, IUPAC codes and windows that hang over the record's ends included); eval mode refuses the
    symbol form (forward_packed is the packed entry there)."""
    from mural_amd.data import PackedGenome, SymbolWindows
    fx = U.load("snv_train_T.npz")
    rng = np.random.default_rng(11)
    seq = rng.choice(np.frombuffer(b"ACGTNRYKMSWBDHV", np.uint8), size=6000,
                     p=[.24, .24, .24, .24] + [.04 / 11] * 11).tobytes().decode()
    genome = PackedGenome.from_sequence(seq, "cuda")
    R = (product_from_hp(fx["hp"])[0].seq_len - 1) // 2
    B = 48
    pos = torch.from_numpy(rng.integers(0, len(seq), size=B)).cuda()
    pos[:3] = torch.tensor([0, 5, len(seq) - 1])
    strand = torch.from_numpy(rng.integers(0, 2, size=B).astype(np.uint8)).cuda()
    cat = genome.encode_kmer(pos, strand, int(fx["hp"][0]), int(fx["hp"][1]))
    dense = genome.encode_onehot(pos, strand, R)
    syms = genome.encode_symbols(pos, strand, R)
    assert isinstance(syms, SymbolWindows) and syms.sym.dtype == torch.uint8 and tuple(syms.shape) == (B, 2 * R + 1)
    y = torch.from_numpy(rng.integers(0, 4, size=B)).cuda()
    crit = nn.CrossEntropyLoss(reduction="sum")
    results = []
    for x in (dense, syms):
        model, _ = product_from_hp(fx["hp"])
        model.load_state_dict(U.snv_state_for(fx, U.snv_oracle_from_hp(fx["hp"])))
        model = model.cuda().train()
        torch.manual_seed(5)                                   # the dropout seeds of the step are drawn from torch's generator
        out = model((torch.zeros(B, 1, device="cuda"), cat), x)
        crit(out, y).backward()
        results.append((out.detach().clone(), [None if p.grad is None else p.grad.detach().clone() for p in model.parameters()],
                        [b.detach().clone() for b in model.buffers()]))
    (o1, g1, b1), (o2, g2, b2) = results
    assert torch.equal(o1, o2)
    assert sum(g is not None for g in g1) > 20
    # (the first-layer and embedding gradients are sums of float atomics: equal up to their order between ANY two runs)
    assert all((a is None and b is None) or float((a - b).abs().max()) <= 1e-5 * (float(a.abs().max()) + 1e-6) for a, b in zip(g1, g2))
    assert all(torch.equal(a, b) for a, b in zip(b1, b2))
    model.eval()
    with pytest.raises(TypeError):
        model((torch.zeros(B, 1, device="cuda"), cat), syms)


def test_flat_adam_matches_torch_adam_and_falls_back():
    """mural_amd.train.Adam (one launch over the flat parameter / gradient / moment buffers of the library's training step) against
    torch.optim.Adam(fused=True) stepping a SHADOW copy of the parameters with the very same gradients: the same weights after every
    step (an update is lr-sized = 1e-3, a rounding of it 1e-10; the gradients are shared because a parameter whose true gradient is
    zero -- a conv bias in front of a BatchNorm -- gets rounding noise that Adam turns into a +-lr step, in ANY two runs); the
    one-launch path is the one that ran; a step whose gradient is not the backward's view (replaced by a clone) goes through torch's
    own step on the same state and the next one is one launch again; the state loads into a plain torch.optim.Adam and back; weight
    decay is the L2 form of torch.optim.Adam; eval-mode predictions see the stepped weights."""
    from mural_amd.train import Adam, clip_grad_norm_
    fx = U.load("snv_train_T.npz")
    cat = torch.from_numpy(fx["cat"]).cuda()
    x = U.onehot(fx["codes"]).cuda()
    y = torch.from_numpy(fx["y"]).cuda()
    cont = torch.zeros(len(cat), 1, device="cuda")
    crit = nn.CrossEntropyLoss(reduction="sum")
    for wd in (0.0, 1e-2):
        model, _ = product_from_hp(fx["hp"])
        model.load_state_dict(U.snv_state_for(fx, U.snv_oracle_from_hp(fx["hp"])))
        model = model.cuda().train()
        params = [p for p in model.parameters() if p.numel()]
        shadow = [p.detach().clone().requires_grad_() for p in params]
        names = [k for k, p in model.named_parameters() if p.numel()]
        opt = Adam(model.parameters(), lr=1e-3, weight_decay=wd)
        ref = torch.optim.Adam(shadow, lr=1e-3, weight_decay=wd, fused=True)

        def one_step(o, r, seed, spoil=False):
            torch.manual_seed(seed)
            loss = crit(model((cont, cat), x), y)
            o.zero_grad()
            loss.backward()
            clip_grad_norm_(model, 10)
            if spoil:
                w = model.conv1_2[1].weight
                w.grad = w.grad.clone()
            for p, q in zip(params, shadow):
                q.grad = p.grad.detach().clone()
            o.step()
            r.step()

        def same(what):
            for k, a, b in zip(names, params, shadow):
                d = float((a.detach() - b.detach()).abs().max())
                assert d <= 2e-7, (what, wd, k, d)

        for s in range(4):
            one_step(opt, ref, 10 + s)
            same(("flat", s))
        assert opt._flat is not None and opt._flat_t == 4, "the one-launch step did not run"
        base = opt._flat[2]
        assert all(base.data_ptr() <= p.data_ptr() < base.data_ptr() + 4 * base.numel() for p in params)
        one_step(opt, ref, 20, spoil=True)      # a gradient that is not the backward's view: torch's own step on the same state
        assert opt._flat is None
        same("through torch")
        one_step(opt, ref, 21)
        assert opt._flat is not None and opt._flat_t == 6
        same("flat again")
        taken = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=wd, fused=True)      # the state is torch's
        taken.load_state_dict(opt.state_dict())
        assert all(float(st["step"]) == 6.0 for st in taken.state.values())
        one_step(taken, ref, 22)
        same("taken over by torch")
        back = Adam(model.parameters(), lr=1e-3, weight_decay=wd)
        back.load_state_dict(taken.state_dict())
        one_step(back, ref, 23)
        assert back._flat is not None and back._flat_t == 8
        same("taken back")
    # eval-mode predictions see the stepped weights (the kernel bumps no tensor version): a step taken while the model is in eval mode
    torch.manual_seed(24)
    loss = crit(model((cont, cat), x), y)
    back.zero_grad()
    loss.backward()
    model.eval()
    with torch.no_grad():
        before = model((cont, cat), x).clone()      # (builds the folded copy of the weights)
    back.step()
    assert back._flat is not None and back._flat_t == 9
    with torch.no_grad():
        after = model((cont, cat), x)
    twin, _ = product_from_hp(fx["hp"])
    twin.load_state_dict({k: v.detach().clone() for k, v in model.state_dict().items()})
    twin = twin.cuda().eval()
    with torch.no_grad():
        want = twin((cont, cat), x)
    assert not torch.equal(before, after) and float((after - want).abs().max()) <= 1e-5


def test_fused_cross_entropy_sum_matches_torch():
    """mural_amd.train.CrossEntropySum against torch.nn.CrossEntropyLoss(reduction='sum'): value within the rounding of a 4096-term
    float sum, gradient within 1e-6, bitwise reproducible, upstream gradients other than 1 honoured, rows with large logits stable;
    tensors it does not take (float64, more than 65536 rows) go to torch's implementation."""
    from mural_amd.train import CrossEntropySum
    g = torch.Generator().manual_seed(2)
    for B, nc, scale in ((4096, 4, 1.0), (77, 8, 30.0), (1, 3, 1.0), (5000, 2, 0.1), (5001, 4, 20.0), (3, 4, 1.0)):
        x0 = (torch.randn(B, nc, generator=g) * scale).cuda()
        y = torch.randint(0, nc, (B,), generator=g).cuda()
        xa, xb = x0.clone().requires_grad_(), x0.clone().requires_grad_()
        la = CrossEntropySum()(xa, y)
        lb = nn.CrossEntropyLoss(reduction="sum")(xb, y)
        (la * 0.5).backward()
        (lb * 0.5).backward()
        assert abs(la.item() - lb.item()) <= 2e-6 * abs(lb.item()) + 1e-6, (B, nc, la.item(), lb.item())
        assert float((xa.grad - xb.grad).abs().max()) <= 1e-6
        xc = x0.clone().requires_grad_()
        lc = CrossEntropySum()(xc, y)
        assert torch.equal(lc, la)
    x64 = torch.randn(9, 4, dtype=torch.float64, device="cuda")
    y9 = torch.randint(0, 4, (9,), device="cuda")
    assert torch.allclose(CrossEntropySum()(x64, y9), nn.CrossEntropyLoss(reduction="sum")(x64, y9))
