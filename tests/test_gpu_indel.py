"""GPU parity: HIP UNet_Small forward vs the reference's golden vectors (relative tolerance 1e-4 on the positive
Softplus scores, 1e-5 abs on the softmax probabilities callers derive from them)."""
import glob
import os

import numpy as np
import pytest
import torch

from tests import _util as U

pytestmark = pytest.mark.gpu
INDEL_FORWARD = sorted(os.path.basename(p) for p in glob.glob(os.path.join(U.GOLDEN, "indel_*.npz"))
                       if not os.path.basename(p).startswith("indel_train_"))


def product_from(fx):
    from mural_amd.model import model_choice
    R, C, k, n_class, rev = [int(v) for v in fx["hp"]]
    cfg = dict(CNN_out_channels=C, CNN_kernel_size=k, down_list=[int(d) for d in fx["down"]], use_reverse=bool(rev))
    return model_choice(0, cfg, dict(n_class=n_class), "indel")


@pytest.mark.parametrize("name", INDEL_FORWARD)
def test_forward_matches_reference(name):
    fx = U.load(name)
    model = product_from(fx)
    orc = U.indel_oracle_from_hp(fx["hp"], fx["down"])
    assert list(model.state_dict().keys()) == list(orc.state_dict().keys())
    model.load_state_dict(U.indel_state_for(fx, orc), strict=True)
    model = model.cuda().eval()
    with torch.no_grad():
        out = model(U.onehot(fx["codes"]).cuda()).cpu().numpy()
    want = fx["out"]
    assert out.shape == want.shape
    assert np.abs(out - want).max() <= 1e-4 * max(1.0, np.abs(want).max()), name
    sm = lambda a: np.exp(a - a.max(1, keepdims=True)) / np.exp(a - a.max(1, keepdims=True)).sum(1, keepdims=True)
    assert np.abs(sm(out.astype(np.float64)) - sm(want.astype(np.float64))).max() <= 1e-5


def test_batch_larger_than_chunk_and_packed_path(monkeypatch):
    from mural_amd.data import PackedGenome
    monkeypatch.setenv("MURAL_INDEL_CHUNK", "256")      # 300 positions: two chunks, one per stream (the default chunk is 4096)
    from oracle import encode_ref
    fx = U.load("indel_synth_small.npz")
    model = product_from(fx)
    orc = U.indel_oracle_from_hp(fx["hp"], fx["down"])
    sd = U.indel_state_for(fx, orc)
    model.load_state_dict(sd)
    orc.load_state_dict(sd)
    model = model.cuda().eval()
    orc.eval()
    rng = np.random.default_rng(3)
    seq = rng.choice(np.frombuffer(b"ACGTN", np.uint8), size=30000, p=[.247, .247, .247, .247, .012]).tobytes().decode()
    codes = encode_ref.seq_to_codes(seq)
    pos = rng.integers(0, len(seq), size=300)
    pos[:3] = [0, 2, len(seq) - 1]
    strand = rng.integers(0, 2, size=300).astype(np.uint8)
    sym = ["-" if s else "+" for s in strand]
    R = int(fx["hp"][0])
    x = torch.from_numpy(encode_ref.onehot_encode(codes, pos, sym, R, "indel"))
    with torch.no_grad():
        want = orc(x).numpy()
    genome = PackedGenome.from_sequence(seq, "cuda")
    got = model.forward_packed(genome, torch.from_numpy(pos).cuda(), torch.from_numpy(strand).cuda(), R).cpu().numpy()
    assert np.abs(got - want).max() <= 1e-4 * max(1.0, np.abs(want).max())


def test_forward_does_not_read_unwritten_workspace(monkeypatch):
    """The eval-mode forward's workspace (two chunks in flight) with 0xFF-poisoned allocations: same scores."""
    from mural_amd.model import model_indel as MI
    from tests.test_gpu_snv import _PoisonedTorch
    monkeypatch.setenv("MURAL_INDEL_CHUNK", "512")      # 700 windows: two chunks in flight
    fx = U.load("indel_synth_small.npz")
    model = product_from(fx)
    orc = U.indel_oracle_from_hp(fx["hp"], fx["down"])
    model.load_state_dict(U.indel_state_for(fx, orc))
    model = model.cuda().eval()
    x = U.onehot(np.random.default_rng(5).integers(0, 4, size=(700, fx["codes"].shape[1])).astype(np.uint8)).cuda()
    with torch.no_grad():
        want = model(x).cpu().numpy()
        monkeypatch.setattr(MI, "torch", _PoisonedTorch())
        model._ws = None
        got = model(x).cpu().numpy()
    assert np.isfinite(got).all() and np.abs(got - want).max() <= 1e-6 * max(1.0, np.abs(want).max())


def test_forward_writes_stay_inside_their_workspace_regions(monkeypatch):
    """The eval-mode forward's workspace (both lanes) with 4 KB of poisoned guard bytes behind every region: all survive the call."""
    import ctypes as C
    from mural_amd import _lib
    from mural_amd.model import model_indel as MI
    from tests.test_gpu_snv import _PoisonedTorch
    guard = 4096
    monkeypatch.setenv("MURAL_DEBUG_WS_GUARD", str(guard))
    monkeypatch.setenv("MURAL_INDEL_CHUNK", "2048")     # 2500 windows: both lanes
    monkeypatch.setattr(MI, "torch", _PoisonedTorch())
    fx = U.load("indel_synth_small.npz")
    model = product_from(fx)
    orc = U.indel_oracle_from_hp(fx["hp"], fx["down"])
    model.load_state_dict(U.indel_state_for(fx, orc))
    model = model.cuda().eval()
    x = U.onehot(np.random.default_rng(6).integers(0, 4, size=(2500, fx["codes"].shape[1])).astype(np.uint8)).cuda()
    model._ws = None
    with torch.no_grad():
        out = model(x)
    assert torch.isfinite(out).all()
    layout = (C.c_size_t * 128)()
    n_regions = _lib.lib().mural_debug_last_ws_layout(layout, 64)
    assert n_regions >= 16          # two lanes
    ws = model._ws.cpu().numpy()
    for i in range(n_regions):
        off, size = layout[2 * i], layout[2 * i + 1]
        zone = ws[off + size:off + size + guard]
        assert len(zone) == guard and (zone == 255).all(), f"region {i}: a kernel wrote behind its {size} bytes"


def test_incompatible_length_is_rejected():
    fx = U.load("indel_synth_small.npz")
    model = product_from(fx).cuda().eval()
    with pytest.raises(ValueError):
        model(torch.zeros(2, 4, 800, device="cuda"))


# ------------------------------------------------------------------------------------------------ training mode
def _train_step(model, x, y):
    preds = model(x)
    loss = torch.nn.CrossEntropyLoss(reduction="sum")(preds, y)
    model.zero_grad()
    loss.backward()
    return preds, loss


@pytest.mark.parametrize("tag", ["rev", "norev"])
def test_train_step_matches_reference(tag):
    """One training step (batch-statistics BatchNorm, CE(sum) loss, backward) against the reference's own step (G14):
    scores 1e-5 of their scale, loss 1e-5 relative, every gradient 2e-4 of (its tensor's max + 1e-2), running statistics 1e-5."""
    fx = U.load(f"indel_train_{tag}.npz")
    model = product_from(fx)
    orc = U.indel_oracle_from_hp(fx["hp"], fx["down"])
    model.load_state_dict(U.indel_state_for(fx, orc), strict=True)
    model = model.cuda().train()
    model.out_fc[1].p = 0.0
    preds, loss = _train_step(model, U.onehot(fx["codes"]).cuda(), torch.from_numpy(fx["y"]).cuda())
    # Softplus scores of magnitude 1..3: 1e-5 of the score scale (the 1e-5 absolute bar is on the probabilities derived from them)
    assert np.abs(preds.detach().cpu().numpy() - fx["preds"]).max() <= 1e-5 * max(1.0, float(np.abs(fx["preds"]).max()))
    assert abs(loss.item() - float(fx["loss"])) <= 1e-5 * abs(float(fx["loss"]))
    grads = {k: p.grad.cpu().numpy() for k, p in model.named_parameters()}
    for k, got in grads.items():
        want = fx["g::" + k]
        diff = np.abs(got - want).max()
        tol = 2e-4 * (np.abs(want).max() + 1e-2)          # the bar of the SNV step (test_gpu_train.py)
        if k.endswith(".bias") and k[:-4] + "weight" in grads and want.ndim == 1 and fx["g::" + k[:-4] + "weight"].ndim == 3:
            # a conv bias gradient is the plain sum of the dy whose products with x form dW; in front of a batch-statistics
            # BatchNorm it is mathematically zero, and behind the 5-row BatchNorm of out_fc it cancels to ~1e-3 of dW: its
            # fp32 error follows the scale of dW (the reference's own fp32 value is 2e-5 away from a float64 evaluation)
            tol = max(tol, 2e-5 * np.abs(fx["g::" + k[:-4] + "weight"]).max())
        assert diff <= tol, (k, diff, tol)
    for k, b in model.named_buffers():
        assert np.abs(b.cpu().numpy().astype(np.float64) - fx["b::" + k]).max() <= 1e-5, k
    gnorm = torch.nn.utils.clip_grad_norm_(model.parameters(), 1e9)
    assert abs(float(gnorm) - float(fx["gnorm"])) <= 1e-4 * float(fx["gnorm"])


def test_train_general_conv_against_torch():
    """The general Conv1d op (stride, padding, upsampled input) and its backward against torch's conv1d on the shapes of the
    U-Net and a few odd ones (ragged lengths, tiny rows)."""
    from mural_amd.model.indel_train import Conv
    rng = torch.Generator().manual_seed(5)
    cases = [(3, 4, 8, 7, 1, 3, 1, 500), (2, 8, 16, 7, 4, 3, 1, 501), (2, 32, 40, 7, 5, 3, 1, 77), (5, 40, 48, 7, 2, 3, 1, 4),
             (2, 48, 40, 7, 1, 3, 2, 9), (2, 16, 8, 7, 1, 3, 4, 130), (3, 48, 96, 5, 1, 2, 1, 70), (3, 96, 48, 1, 1, 0, 1, 70),
             (1, 4, 4, 7, 1, 3, 1, 2000)]
    for B, Cin, Cout, K, stride, pad, up, L in cases:
        x = torch.randn((B, Cin, L), generator=rng)
        w = torch.randn((Cout, Cin, K), generator=rng) / (Cin * K) ** 0.5
        b = torch.randn(Cout, generator=rng)
        xr, wr, br = x.clone().requires_grad_(), w.clone().requires_grad_(), b.clone().requires_grad_()
        xu = xr.repeat_interleave(up, dim=2) if up > 1 else xr
        yr = torch.nn.functional.conv1d(xu, wr, br, stride=stride, padding=pad)
        g = torch.randn(yr.shape, generator=rng)
        yr.backward(g)
        xd, wd, bd = x.cuda().requires_grad_(), w.cuda().requires_grad_(), b.cuda().requires_grad_()
        yd = Conv.apply(xd, wd, bd, stride, pad, up)
        yd.backward(g.cuda())
        case = (B, Cin, Cout, K, stride, pad, up, L)
        assert yd.shape == yr.shape, case
        assert (yd.detach().cpu() - yr.detach()).abs().max() <= 2e-5 * max(1.0, float(yr.abs().max())), case
        for got, want, name in ((xd.grad, xr.grad, "dx"), (wd.grad, wr.grad, "dW"), (bd.grad, br.grad, "db")):
            assert (got.cpu() - want).abs().max() <= 1e-4 * max(1.0, float(want.abs().max())), (case, name)


def test_conv_bn_unit_against_torch():
    """The U-Net's training unit as one autograd node (indel_train.ConvBn: Conv1d -> batch-statistics BatchNorm -> act -> residual
    adds, ``mural_op_convg_bn_fwd / _bwd``) against the same chain of torch modules in float64: outputs, every gradient (input,
    conv weight / bias, BatchNorm weight / bias, both residuals) and the running statistics, over strides, upsampling factors, the
    three activations and batches that do not fill a tile."""
    from mural_amd.model.indel_train import ConvBn
    rng = torch.Generator().manual_seed(11)
    cases = [(3, 8, 16, 5, 1, 2, 1, 300, 2, True), (2, 16, 8, 1, 1, 0, 1, 257, 0, True), (2, 8, 16, 7, 4, 3, 1, 403, 0, False),
             (4, 40, 48, 7, 2, 3, 1, 16, 1, False), (3, 48, 40, 7, 1, 3, 2, 8, 0, False), (2, 16, 8, 7, 1, 3, 4, 50, 0, True),
             (5, 24, 48, 5, 1, 2, 1, 80, 2, False), (2, 32, 24, 7, 1, 3, 5, 16, 0, True)]
    for B, Cin, Cout, K, stride, pad, up, L, act, with_res in cases:
        x = torch.randn((B, Cin, L), generator=rng)
        conv = torch.nn.Conv1d(Cin, Cout, K, stride=stride, padding=pad)
        bn = torch.nn.BatchNorm1d(Cout)
        with torch.no_grad():
            conv.weight.copy_(torch.randn(conv.weight.shape, generator=rng) / (Cin * K) ** 0.5)
            conv.bias.copy_(torch.randn(Cout, generator=rng))
            bn.weight.copy_(1 + 0.3 * torch.randn(Cout, generator=rng))
            bn.bias.copy_(0.3 * torch.randn(Cout, generator=rng))
        ref_c, ref_b = torch.nn.Conv1d(Cin, Cout, K, stride=stride, padding=pad).double(), torch.nn.BatchNorm1d(Cout).double()
        ref_c.load_state_dict({k: v.double() for k, v in conv.state_dict().items()})
        ref_b.load_state_dict({k: (v.double() if v.is_floating_point() else v) for k, v in bn.state_dict().items()})
        xr = x.double().requires_grad_()
        xu = xr.repeat_interleave(up, dim=2) if up > 1 else xr
        u = ref_b(ref_c(xu))
        zr = [u, torch.relu(u), torch.nn.functional.silu(u)][act]
        r1 = torch.randn(zr.shape, generator=rng) if with_res else None
        r2 = torch.randn(zr.shape, generator=rng) if with_res else None
        r1r = r1.double().requires_grad_() if with_res else None
        r2r = r2.double().requires_grad_() if with_res else None
        if with_res:
            zr = zr + r1r + r2r
        g = torch.randn(zr.shape, generator=rng)
        zr.backward(g.double())
        conv, bn = conv.cuda().train(), bn.cuda().train()
        xd = x.cuda().requires_grad_()
        r1d = r1.cuda().requires_grad_() if with_res else None
        r2d = r2.cuda().requires_grad_() if with_res else None
        zd = ConvBn.apply(xd, conv.weight, conv.bias, bn.weight, bn.bias, bn, stride, pad, up, act, r1d, r2d)
        zd.backward(g.cuda())
        case = (B, Cin, Cout, K, stride, pad, up, L, act, with_res)
        close = lambda got, want, tol: float((got.detach().cpu().double() - want.detach()).abs().max()) <= tol * max(1.0, float(want.abs().max()))   # noqa: E731
        assert close(zd, zr, 2e-5), case
        assert close(xd.grad, xr.grad, 1e-4), (case, "dx")
        assert close(conv.weight.grad, ref_c.weight.grad, 1e-4), (case, "dW")
        assert close(bn.weight.grad, ref_b.weight.grad, 1e-4) and close(bn.bias.grad, ref_b.bias.grad, 1e-4), (case, "BatchNorm")
        # (the conv bias in front of a batch-statistics BatchNorm has a mathematically zero gradient: only its scale is checked)
        assert float(conv.bias.grad.abs().max()) <= 1e-3 * max(1.0, float(ref_c.weight.grad.abs().max())), (case, "db")
        if with_res:
            assert close(r1d.grad, r1r.grad, 1e-6) and close(r2d.grad, r2r.grad, 1e-6), (case, "residuals")
        assert close(bn.running_mean, ref_b.running_mean, 1e-5) and close(bn.running_var, ref_b.running_var, 1e-5), (case, "running")


def test_train_step_does_not_read_unwritten_scratch(monkeypatch):
    """Every buffer the training step allocates with torch.empty / empty_like (conv outputs, BatchNorm states, gradients, layouts,
    partial sums) must be completely written before a kernel reads it: the same step with those allocations pre-filled with NaN
    gives the same loss and gradients.  (The per-unit autograd composition of model/indel_train.py, which the generic SNV towers
    share; the one-call step has its own workspace tests below.)"""
    from mural_amd.model import indel_train as IT
    monkeypatch.setenv("MURAL_INDEL_TRAIN_PER_UNIT", "1")
    fx = U.load("indel_train_rev.npz")
    model = product_from(fx)
    orc = U.indel_oracle_from_hp(fx["hp"], fx["down"])
    model.load_state_dict(U.indel_state_for(fx, orc), strict=True)
    model = model.cuda().train()
    model.out_fc[1].p = 0.0
    x, y = U.onehot(fx["codes"]).cuda(), torch.from_numpy(fx["y"]).cuda()
    _, loss0 = _train_step(model, x, y)
    ref = {k: p.grad.clone() for k, p in model.named_parameters()}

    class NanTorch:
        def __getattr__(self, k):
            return getattr(torch, k)

        @staticmethod
        def empty(*a, **k):
            t = torch.empty(*a, **k)
            return t.fill_(float("nan")) if t.is_floating_point() else t

        @staticmethod
        def empty_like(*a, **k):
            t = torch.empty_like(*a, **k)
            return t.fill_(float("nan")) if t.is_floating_point() else t

    monkeypatch.setattr(IT, "torch", NanTorch())
    monkeypatch.setattr(IT, "_scratch", {})
    _, loss1 = _train_step(model, x, y)
    assert abs(loss1.item() - loss0.item()) <= 1e-5 * abs(loss0.item())
    for k, p in model.named_parameters():
        assert torch.isfinite(p.grad).all(), k
        assert float((p.grad - ref[k]).abs().max()) <= 1e-4 * (float(ref[k].abs().max()) + 1e-3), k


def test_train_step_writes_stay_inside_their_buffers(monkeypatch):
    """Every tensor the training step allocates is carved out of one arena with 4 KB guard zones (0xAB) around it: after a forward +
    backward every guard byte is intact -- no kernel of the step writes outside the buffer it was given.  (Per-unit composition.)"""
    from mural_amd.model import indel_train as IT
    monkeypatch.setenv("MURAL_INDEL_TRAIN_PER_UNIT", "1")
    fx = U.load("indel_train_rev.npz")
    model = product_from(fx)
    orc = U.indel_oracle_from_hp(fx["hp"], fx["down"])
    model.load_state_dict(U.indel_state_for(fx, orc), strict=True)
    model = model.cuda().train()
    x, y = U.onehot(fx["codes"]).cuda(), torch.from_numpy(fx["y"]).cuda()
    guard = 4096
    arena = torch.full((1 << 30,), 0xAB, dtype=torch.uint8, device="cuda")
    allocs, top = [], [0]

    def carve(shape, dtype):
        n = 1
        for d in shape:
            n *= int(d)
        nbytes = n * torch.empty(0, dtype=dtype).element_size()
        off = (top[0] + guard + 511) // 512 * 512
        assert off + nbytes + guard <= arena.numel(), "test arena too small"
        allocs.append((off, nbytes, tuple(shape)))
        top[0] = off + nbytes
        return arena[off:off + nbytes].view(dtype).view(tuple(int(d) for d in shape))

    class GuardTorch:
        def __getattr__(self, k):
            return getattr(torch, k)

        @staticmethod
        def empty(*a, **k):
            shape = a[0] if len(a) == 1 and isinstance(a[0], (tuple, list, torch.Size)) else a
            return carve((shape,) if isinstance(shape, int) else tuple(shape), k.get("dtype", torch.float32))

        @staticmethod
        def empty_like(t, **k):
            return carve(tuple(t.shape), t.dtype)

    monkeypatch.setattr(IT, "torch", GuardTorch())
    monkeypatch.setattr(IT, "_scratch", {})
    _train_step(model, x, y)
    torch.cuda.synchronize()
    host = arena[:top[0] + guard].cpu()
    assert len(allocs) > 200
    end = 0
    for off, nbytes, shape in allocs:
        assert bool((host[end:off] == 0xAB).all()), f"a kernel wrote outside its buffer next to a tensor of shape {shape}"
        end = off + nbytes
    assert bool((host[end:end + guard] == 0xAB).all())


def test_clip_grad_norm_on_separate_gradient_tensors_matches_torch(monkeypatch):
    """mural_amd.train.clip_grad_norm_ on a model whose gradients are separate tensors (UNet_Small on the per-unit composition) ==
    torch.nn.utils.clip_grad_norm_: same total norm, same clipped gradients, and no clipping below the bound."""
    from mural_amd.train import clip_grad_norm_
    monkeypatch.setenv("MURAL_INDEL_TRAIN_PER_UNIT", "1")
    fx = U.load("indel_train_rev.npz")
    model = product_from(fx)
    orc = U.indel_oracle_from_hp(fx["hp"], fx["down"])
    model.load_state_dict(U.indel_state_for(fx, orc), strict=True)
    model = model.cuda().train()
    model.out_fc[1].p = 0.0
    x, y = U.onehot(fx["codes"]).cuda(), torch.from_numpy(fx["y"]).cuda()
    _train_step(model, x, y)
    g0 = [p.grad.clone() for p in model.parameters()]
    for bound in (1e-3, 1e9):
        for p, g in zip(model.parameters(), g0):
            p.grad = g.clone()
        want = torch.nn.utils.clip_grad_norm_(model.parameters(), bound)
        ref = [p.grad.clone() for p in model.parameters()]
        for p, g in zip(model.parameters(), g0):
            p.grad = g.clone()
        got = clip_grad_norm_(model, bound)
        assert abs(float(got) - float(want)) <= 1e-6 * float(want)
        for p, r in zip(model.parameters(), ref):
            assert torch.equal(p.grad, r)


def test_train_mode_updates_and_eval_after_training():
    """A few Adam steps in training mode lower the loss, BatchNorm counters advance like nn.BatchNorm1d (the strand-symmetry
    BatchNorm twice per forward), and the eval-mode fused program picks up the updated weights."""
    fx = U.load("indel_train_rev.npz")
    model = product_from(fx)
    orc = U.indel_oracle_from_hp(fx["hp"], fx["down"])
    sd = U.indel_state_for(fx, orc)
    model.load_state_dict(sd, strict=True)
    model = model.cuda().train()
    model.out_fc[1].p = 0.0
    x, y = U.onehot(fx["codes"]).cuda(), torch.from_numpy(fx["y"]).cuda()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    losses = []
    for _ in range(5):
        _, loss = _train_step(model, x, y)
        torch.nn.utils.clip_grad_norm_(model.parameters(), 10)
        opt.step()
        losses.append(loss.item())
    assert losses[-1] < losses[0]
    assert int(model.conv[1].num_batches_tracked) == 7 + 10 and int(model.out_fc[0].num_batches_tracked) == 7 + 5
    model.eval()
    orc.load_state_dict({k: v.cpu() for k, v in model.state_dict().items()})
    orc.eval()
    with torch.no_grad():
        got = model(x).cpu().numpy()
        want = orc(U.onehot(fx["codes"])).numpy()
    assert np.abs(got - want).max() <= 1e-4 * max(1.0, np.abs(want).max())


def test_graphed_indel_train_step_matches_eager():
    """hipGraph replay of the whole UNet_Small step (mural_amd.train.GraphedIndelTrainStep) == the eager step: same loss and
    parameters after the same batches (dropout off, plain SGD: see test_gpu_train.test_graphed_train_step_matches_eager), BatchNorm
    counters advanced by the replays, and the eval-mode program picks up the replayed weights."""
    from mural_amd.train import GraphedIndelTrainStep
    fx = U.load("indel_train_rev.npz")
    orc = U.indel_oracle_from_hp(fx["hp"], fx["down"])
    x, y = U.onehot(fx["codes"]).cuda(), torch.from_numpy(fx["y"]).cuda()
    crit = torch.nn.CrossEntropyLoss(reduction="sum")

    def make():
        model = product_from(fx)
        model.load_state_dict(U.indel_state_for(fx, orc), strict=True)
        model = model.cuda().train()
        model.out_fc[1].p = 0.0
        return model, torch.optim.SGD(model.parameters(), lr=1e-3, momentum=0.9)

    eager, opt_e = make()
    for _ in range(3 + 2):     # GraphedIndelTrainStep: 3 eager warm-up steps + 2 replays
        _, loss = _train_step(eager, x, y)
        torch.nn.utils.clip_grad_norm_(eager.parameters(), 10)
        opt_e.step()
    graphed, opt_g = make()
    step = GraphedIndelTrainStep(graphed, opt_g, crit, x, y)
    for _ in range(2):
        loss_g = step(x, y)
    step.finish()
    assert abs(loss_g.item() - loss.item()) <= 1e-3 * abs(loss.item())
    for (k, p), (_, q) in zip(eager.named_parameters(), graphed.named_parameters()):
        assert float((p.detach() - q.detach()).abs().max()) <= 2e-4 * (float(p.detach().abs().max()) + 1e-3), k
    assert int(graphed.conv[1].num_batches_tracked) == int(eager.conv[1].num_batches_tracked)
    graphed.eval()
    eager.eval()
    with torch.no_grad():
        a, b = graphed(x).cpu().numpy(), eager(x).cpu().numpy()
    assert np.abs(a - b).max() <= 1e-4 * max(1.0, np.abs(b).max())


def test_generic_conv1d_kernels_match_torch_fp64():
    """Both engines of the generic Conv1d behind the U-Net (vector ALU, csrc/conv1d.hip; MFMA implicit GEMM, csrc/conv1d_mfma.hip)
    against torch in float64 over the layer geometries of UNet_Small: strides 4 / 5 / 2, upsampling 2 / 5, k = 7 / 5 / 1,
    16..96 channels (incl. 24 / 40: half-empty MFMA blocks), rows of 8..2000 columns, SiLU / ReLU / Softplus, residuals, and a
    batch that does not fill the last tile."""
    import os
    import subprocess
    import sys
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "gpu_debug_conv1d.py")
    env = {k: v for k, v in os.environ.items() if k != "MURAL_TEST_VERBOSE"}
    out = subprocess.run([sys.executable, tool], capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert "worst" in out.stdout


def test_mfma_weight_gradient_matches_torch_fp64():
    """The weight / bias gradient of the general Conv1d on the matrix cores (csrc/conv_wgrad_mfma.hip: interior groups with both
    operands straight from global memory, gathered edge segments) against torch's conv gradients in float64: the U-Net's layer
    geometries (strides 4 / 5 / 2, upsampling 2 / 4, k = 7 / 5 / 3 / 1, 4..96 channels), rows of 4..8000 columns -- rows that are
    all edge, rows with leftover segments, ragged rows -- and batches that do not fill the last group."""
    import os
    import subprocess
    import sys
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "gpu_debug_wgrad.py")
    env = {k: v for k, v in os.environ.items() if k not in ("TIME", "MURAL_WGRAD_MFMA")}
    out = subprocess.run([sys.executable, tool], capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert "worst" in out.stdout


def test_fused_convblock_forms_match_torch_fp64():
    """One fused ConvBlock launch (mural_debug_convblock): plain / with the k = 7 front (upsampled or not) / skip tensor / tail, at 8
    channels in BOTH forms (vector ALU; split: the convs on the matrix cores) and at 16 channels, rows of 37..8000 columns, against
    float64 (tools/gpu_debug_convblock.py: 256 cases, relative error <= 3e-6, every CU's LDS filled with NaN in front of each launch)."""
    import os
    import subprocess
    import sys
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "gpu_debug_convblock.py")
    env = {k: v for k, v in os.environ.items() if k not in ("TIME", "VERBOSE", "MURAL_CONVBLOCK8_VALU", "MURAL_CONVBLOCK_DIRECT")}
    # twice: the library's routing (small launches of the 16 / 24-channel block on the LDS-tiled kernel), then its barrier-free form
    # forced for every size (rows that are all edge segments, ragged rows)
    # (and once with the persistent level-0 kernels of indel_level0.hip off: the split form's own polyphase / genome-fed fronts)
    # (and the 32-channel block on rows of 1 .. 80 columns, one row per workgroup pass: convblock_deep.hip, DEEP=1 selects those cases)
    for extra in ({}, {"MURAL_CONVBLOCK_DIRECT": "2"}, {"MURAL_INDEL_DEC0": "0", "MURAL_INDEL_ENC0": "0"}, {"DEEP": "1"}):
        out = subprocess.run([sys.executable, tool], capture_output=True, text=True, timeout=900, env={**env, **extra})
        assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
        assert "worst" in out.stdout


@pytest.mark.parametrize("L", [16000, 64000])
def test_long_windows_match_oracle(L):
    """The reference advertises INDEL inputs of up to 64 kb (CHANGELOG:13): the human-insertion geometry at L = 16000 and 64000
    (synthetic weights, use_reverse) through the dense and the packed entry against the oracle."""
    from mural_amd.data import PackedGenome
    from mural_amd.model import model_choice
    from oracle import encode_ref, indel_ref, synth
    down = [1, 4, 5, 5, 5, 2]
    orc = indel_ref.build(n_class=8, channels=8, ksize=7, down_list=down, use_reverse=True)
    sd = synth.synth_state_dict(orc.state_dict(), 640 + L // 1000)
    orc.load_state_dict(sd)
    orc.eval()
    model = model_choice(0, dict(CNN_out_channels=8, CNN_kernel_size=7, down_list=down, use_reverse=True), dict(n_class=8), "indel")
    model.load_state_dict(sd)
    model = model.cuda().eval()
    rng = np.random.default_rng(L)
    n = 3 * L
    seq = rng.choice(np.frombuffer(b"ACGTN", np.uint8), size=n, p=[.248, .248, .248, .248, .008]).tobytes().decode()
    codes = encode_ref.seq_to_codes(seq)
    pos = np.array([0, L // 3, n // 2, n - 1, n - L // 2])
    strand = np.array([0, 1, 0, 1, 1], np.uint8)
    sym = ["-" if v else "+" for v in strand]
    x = torch.from_numpy(encode_ref.onehot_encode(codes, pos, sym, L // 2, "indel"))
    assert x.shape == (5, 4, L)
    with torch.no_grad():
        want = orc(x).numpy()
        dense = model(x.cuda()).cpu().numpy()
        genome = PackedGenome.from_sequence(seq, "cuda")
        packed = model.forward_packed(genome, torch.from_numpy(pos).cuda(), torch.from_numpy(strand).cuda(), L // 2).cpu().numpy()
    tol = 1e-4 * max(1.0, np.abs(want).max())
    assert np.abs(dense - want).max() <= tol and np.abs(packed - want).max() <= tol


@pytest.mark.parametrize("name", ["indel_pretrained_human_insertion.npz", "indel_synth_small.npz", "indel_synth_c2.npz",
                                  "indel_pretrained_arabidopsis_insertion.npz"])
def test_packed_entry_decodes_inside_the_first_level(name, monkeypatch):
    """mural_indel_forward_packed: the window is decoded from the packed genome inside the first level's kernel and the input layer
    (strand-symmetrising conv, or none) is evaluated per symbol -- against the oracle fed by the oracle encoder, with N runs, all
    IUPAC codes, both strands and windows that hang over both chromosome ends; the materialising fallback gives the same scores."""
    from mural_amd.data import PackedGenome
    from oracle import encode_ref
    fx = U.load(name)
    model = product_from(fx)
    orc = U.indel_oracle_from_hp(fx["hp"], fx["down"])
    sd = U.indel_state_for(fx, orc)
    model.load_state_dict(sd)
    orc.load_state_dict(sd)
    model = model.cuda().eval()
    orc.eval()
    R = int(fx["hp"][0])
    rng = np.random.default_rng(R)
    n = 6 * R + 1000
    alphabet, p = b"ACGTNRYMSWKBDHV", [.2455] * 4 + [.008] + [.001] * 10
    raw = rng.choice(np.frombuffer(alphabet, np.uint8), size=n, p=np.array(p) / sum(p))
    raw[n // 2:n // 2 + 40] = ord("N")
    seq = raw.tobytes().decode()
    codes = encode_ref.seq_to_codes(seq)
    pos = np.r_[[0, 3, R - 1, n - 1, n - R, n // 2 + 7], rng.integers(0, n, size=26)]
    strand = (np.arange(len(pos)) % 2).astype(np.uint8)
    sym = ["-" if v else "+" for v in strand]
    x = torch.from_numpy(encode_ref.onehot_encode(codes, pos, sym, R, "indel"))
    genome = PackedGenome.from_sequence(seq, "cuda")
    tp, ts = torch.from_numpy(pos).cuda(), torch.from_numpy(strand).cuda()
    with torch.no_grad():
        want = orc(x).numpy()
        got = model.forward_packed(genome, tp, ts, R).cpu().numpy()
        monkeypatch.setenv("MURAL_DEBUG_INDEL_NO_GENOME_FRONT", "1")
        slow = model.forward_packed(genome, tp, ts, R).cpu().numpy()
        onehot = genome.encode_onehot(tp, ts, R, "indel")
        dense = model(onehot).cpu().numpy()
        monkeypatch.setenv("MURAL_INDEL_DENSE_SYMBOLS", "0")
        dense_own = model(onehot).cpu().numpy()
    tol = 1e-4 * max(1.0, np.abs(want).max())
    assert np.abs(got - want).max() <= tol, np.abs(got - want).max()
    # the dense entry classifies the one-hot windows into symbol bytes and feeds the same first-level kernel: the same symbols, the same
    # arithmetic as the packed entry (where that kernel serves the geometry); with the switch off it is the materialising fallback's path
    assert np.abs(slow - want).max() <= tol and np.array_equal(slow, dense_own)
    assert np.abs(dense - want).max() <= tol
    _, C_, k_, _, _ = [int(v) for v in fx["hp"]]
    if int(fx["down"][0]) == 1 and C_ == 8 and k_ == 7 and (2 * R) % 4 == 0:      # (the geometry csrc/indel_level0.hip serves)
        assert np.array_equal(dense, got)
    assert model.forward_packed(genome, tp[:0], ts[:0], R).shape == (0, model.n_class)


def test_level0_kernels_and_wide_stores_against_their_fallbacks(monkeypatch):
    """The persistent level-0 launches (csrc/indel_level0.hip: composed 13-tap table front, strided conv of the next level emitted
    by the same launch, decoder with the tail's maximum carried across tiles) and the polyphase up-conv's 16-byte stores through a
    wave-private LDS image, each against the launch form it replaces, on the shipped human-insertion checkpoint at L = 8000 with N
    runs and both strands: the wide stores move the same values (bit-identical scores); the strided conv, the table front and the
    persistent decoder sum in another order (1e-5 relative); so does the one-launch 32-channel block of the fourth level."""
    from mural_amd.data import PackedGenome
    fx = U.load("indel_pretrained_human_insertion.npz")
    model = product_from(fx)
    orc = U.indel_oracle_from_hp(fx["hp"], fx["down"])
    model.load_state_dict(U.indel_state_for(fx, orc))
    model = model.cuda().eval()
    R = int(fx["hp"][0])
    rng = np.random.default_rng(11)
    n = 6 * R + 1000
    raw = rng.choice(np.frombuffer(b"ACGTN", np.uint8), size=n, p=[.2485, .2485, .2485, .2485, .006])
    raw[n // 3:n // 3 + 30] = ord("N")
    genome = PackedGenome.from_sequence(raw.tobytes().decode(), "cuda")
    pos = torch.from_numpy(np.r_[[0, 5, n - 1, n // 3 + 4], rng.integers(0, n, size=28)]).cuda()
    strand = (torch.arange(len(pos)) % 2).to(torch.uint8).cuda()

    def run(**env):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        with torch.no_grad():
            out = model.forward_packed(genome, pos, strand, R).cpu().numpy()
        for k in env:
            monkeypatch.delenv(k)
        return out

    base = run()
    assert np.array_equal(base, run(MURAL_DEBUG_POLY_NARROW="1"))
    assert np.array_equal(base, run(MURAL_XCD_SWIZZLE="0"))      # which workgroup takes which segments changes no value (mfma_tile.h: xcd_wave_index)
    same_values = run(MURAL_INDEL_ENC0_DOWN="0")
    assert np.abs(base - same_values).max() <= 1e-5 * max(1.0, np.abs(base).max())      # (another conv engine sums the taps in another order)
    old = run(MURAL_INDEL_ENC0="0", MURAL_INDEL_DEC0="0")
    assert np.abs(base - old).max() <= 1e-5 * max(1.0, np.abs(base).max())
    two_launches = run(MURAL_INDEL_DEEP="0")      # the 32-channel block of the fourth level as two tiled convs (csrc/convblock_deep.hip)
    assert np.abs(base - two_launches).max() <= 1e-5 * max(1.0, np.abs(base).max())
    own_conv = run(MURAL_INDEL_DEEP_FRONT="0")    # ... with its strided conv as a launch of its own
    assert np.abs(base - own_conv).max() <= 1e-5 * max(1.0, np.abs(base).max())


@pytest.mark.parametrize("name", ["indel_pretrained_human_insertion.npz", "indel_synth_small.npz"])
def test_dense_entry_with_columns_that_are_no_symbol(name, monkeypatch):
    """mural_indel_forward_dense takes ANY float tensor: one-hot / IUPAC-fraction columns travel as symbol bytes through the persistent
    table-driven first level, a column that is no MuRaL symbol (here: random floats, scaled one-hot columns, all-zero columns, isolated
    and in runs, at both window ends) is evaluated from its floats inside that launch -- against the oracle and against the
    launch-per-layer path on the dense tensor."""
    from oracle import encode_ref
    fx = U.load(name)
    model = product_from(fx)
    orc = U.indel_oracle_from_hp(fx["hp"], fx["down"])
    sd = U.indel_state_for(fx, orc)
    model.load_state_dict(sd)
    orc.load_state_dict(sd)
    model = model.cuda().eval()
    orc.eval()
    R = int(fx["hp"][0])
    rng = np.random.default_rng(5 + R)
    n = 4 * R + 500
    raw = rng.choice(np.frombuffer(b"ACGTNRY", np.uint8), size=n, p=[.24, .24, .24, .24, .02, .01, .01])
    codes = encode_ref.seq_to_codes(raw.tobytes().decode())
    pos = np.r_[[0, n - 1, n // 2], rng.integers(0, n, size=9)]
    sym = ["-" if i % 2 else "+" for i in range(len(pos))]
    x = encode_ref.onehot_encode(codes, pos, sym, R, "indel").copy()
    L = x.shape[2]
    x[0, :, 0] = rng.standard_normal(4)                      # first column
    x[1, :, L - 1] = rng.standard_normal(4)                  # last column
    x[2, :, 100:140] = rng.standard_normal((4, 40))          # a run
    x[3, :, rng.integers(0, L, size=60)] *= 0.7              # scaled one-hot columns, scattered
    x[4, :, L // 2 - 3:L // 2 + 3] = 0.0                     # all-zero columns
    x[5] = rng.standard_normal((4, L)).astype(np.float32)    # a window without a single symbol
    xt = torch.from_numpy(x.astype(np.float32))
    with torch.no_grad():
        want = orc(xt).numpy()
        got = model(xt.cuda()).cpu().numpy()
        monkeypatch.setenv("MURAL_INDEL_DENSE_SYMBOLS", "0")
        own = model(xt.cuda()).cpu().numpy()
    tol = 1e-4 * max(1.0, np.abs(want).max())
    assert np.abs(got - want).max() <= tol, np.abs(got - want).max()
    assert np.abs(got - own).max() <= 1e-5 * max(1.0, np.abs(own).max())


@pytest.mark.parametrize("tag", ["rev", "norev"])
def test_one_call_train_step_equals_per_unit_composition_and_keeps_to_its_workspace(tag, monkeypatch):
    """mural_indel_train_forward / _backward (csrc/indel_train_step.hip, one C call per direction) against the per-unit autograd
    composition of the same kernels: same scores, loss, gradients and running statistics (G14 pins both to the reference); then the
    same step with its workspace and flat gradient buffer pre-filled with 0xFF bytes (nothing is read before it is written) and with
    4 KB guard zones behind every workspace region (no kernel writes outside the region it was given)."""
    import ctypes as C
    from mural_amd import _lib
    from mural_amd.model import indel_train_step as ITS
    from tests.test_gpu_snv import _PoisonedTorch
    fx = U.load(f"indel_train_{tag}.npz")
    orc = U.indel_oracle_from_hp(fx["hp"], fx["down"])
    sd = U.indel_state_for(fx, orc)
    x, y = U.onehot(fx["codes"]).cuda(), torch.from_numpy(fx["y"]).cuda()

    def fresh():
        m = product_from(fx)
        m.load_state_dict(sd, strict=True)
        m = m.cuda().train()
        m.out_fc[1].p = 0.0
        return m

    monkeypatch.setenv("MURAL_INDEL_TRAIN_PER_UNIT", "1")
    ref = fresh()
    preds0, loss0 = _train_step(ref, x, y)
    monkeypatch.delenv("MURAL_INDEL_TRAIN_PER_UNIT")
    model = fresh()
    preds1, loss1 = _train_step(model, x, y)
    assert model._train_layout.last_flat is not None                      # the one-call backward ran
    assert float((preds1 - preds0).abs().max()) <= 1e-6 * max(1.0, float(preds0.abs().max()))
    assert abs(loss1.item() - loss0.item()) <= 1e-6 * abs(loss0.item())
    for (k, p), q in zip(model.named_parameters(), ref.parameters()):
        assert float((p.grad - q.grad).abs().max()) <= 2e-5 * (float(q.grad.abs().max()) + 1e-3), k
    for (k, b), c in zip(model.named_buffers(), ref.buffers()):
        assert float((b.double() - c.double()).abs().max()) <= 1e-6 * max(1.0, float(c.double().abs().max())), k
    grads = {k: p.grad.clone() for k, p in model.named_parameters()}
    # poisoned allocations
    monkeypatch.setattr(ITS, "torch", _PoisonedTorch())
    model2 = fresh()
    _, loss2 = _train_step(model2, x, y)
    assert abs(loss2.item() - loss1.item()) <= 1e-6 * abs(loss1.item())
    for k, p in model2.named_parameters():
        assert torch.isfinite(p.grad).all(), k
        assert float((p.grad - grads[k]).abs().max()) <= 1e-5 * (float(grads[k].abs().max()) + 1e-3), k
    # guard zones behind every workspace region
    guard = 4096
    monkeypatch.setenv("MURAL_DEBUG_WS_GUARD", str(guard))
    kept = []
    real_empty = torch.empty

    class Keep(_PoisonedTorch):
        @staticmethod
        def empty(*a, **k):
            t = _PoisonedTorch.empty(*a, **k)
            if t.dtype is torch.uint8:
                kept.append(t)
            return t

    monkeypatch.setattr(ITS, "torch", Keep())
    model3 = fresh()
    _train_step(model3, x, y)
    torch.cuda.synchronize()
    layout = (C.c_size_t * 1024)()
    n_regions = _lib.lib().mural_debug_last_ws_layout(layout, 512)
    assert n_regions >= 100 and kept
    ws = kept[-1].cpu().numpy()
    for i in range(n_regions):
        off, size = layout[2 * i], layout[2 * i + 1]
        zone = ws[off + size:off + size + guard]
        assert len(zone) == guard and (zone == 255).all(), f"region {i}: a kernel wrote behind its {size} bytes"
    del real_empty
