"""Autograd glue for the training-mode HIP building blocks (include/mural_hip.h, ``mural_op_*``).

Each ``torch.autograd.Function`` below launches hand-written HIP kernels for its forward and backward; torch only
provides the tensors, the stream and the autograd graph (the reference relies on stock autograd over ``torch.nn`` ops,
MuRaL/training.py:424-427).  BatchNorm running statistics are updated in place by the forward kernels exactly like
``nn.BatchNorm1d`` in training mode (momentum 0.1, unbiased running variance).
"""
import ctypes as C
import os

import torch

from .. import _lib

EPS, MOMENTUM = 1e-5, 0.1
BN_SLOTS = 32          # include/mural_hip.h MURAL_BN_SLOTS


class _ZeroArena:
    """Zero-initialised float64 scratch handed out in slices: one fill kernel per chunk instead of one per accumulator.
    Slices are never reused (a fresh chunk replaces an exhausted one; views keep old chunks alive as long as needed)."""

    CHUNK = 1 << 17     # doubles (1 MiB)

    def __init__(self):
        self.buf, self.off = None, 0

    def take(self, n, device):
        if self.buf is None or self.buf.device != device or self.off + n > self.buf.numel():
            self.buf, self.off = torch.zeros(max(self.CHUNK, n), dtype=torch.float64, device=device), 0
        out = self.buf[self.off:self.off + n]
        self.off += n
        return out


_arena = _ZeroArena()


def reset_zero_arena():
    """Drop the current chunk (graph capture: the fill of the chunks used inside the graph must be part of the graph)."""
    _arena.buf, _arena.off = None, 0


def _bn_acc(Cn, device):
    """Zeroed accumulator block (BN_SLOTS, 2, C) for batch sums (see mural_op_bn_stats)."""
    return _arena.take(BN_SLOTS * 2 * Cn, device).view(BN_SLOTS, 2, Cn)


def _p(t):
    return None if t is None else t.data_ptr()


_fn_cache = {}
_Tensor = torch.Tensor


def _call(name, *args):
    fn = _fn_cache.get(name)
    if fn is None:
        fn = _fn_cache[name] = getattr(_lib.lib(), name)
    rc = fn(*[a.data_ptr() if isinstance(a, _Tensor) else a for a in args])
    if rc:
        _lib.check(rc)


def _stream(t):
    return _lib.current_stream_ptr(t.device)


def _f32(t):
    if t.dtype is torch.float32 and t.is_contiguous():
        return t                      # only its data pointer is used
    return t.detach().to(torch.float32).contiguous()


_bn_touched = []


def _bn_tick(bn):
    """nn.BatchNorm1d increments num_batches_tracked per training forward; the increments of one model forward are applied
    together (flush_bn_ticks) instead of one tiny kernel per layer."""
    _bn_touched.append(bn.num_batches_tracked)


def flush_bn_ticks():
    flush_input_checks()
    if _bn_touched:
        with torch.no_grad():
            count, first = {}, {}
            for t in _bn_touched:
                count[id(t)] = count.get(id(t), 0) + 1
                first.setdefault(id(t), t)
            once = [first[k] for k, c in count.items() if c == 1]
            if once:
                torch._foreach_add_(once, 1)
            for k, c in count.items():    # a BatchNorm applied more than once per forward (UNet_Small's strand-symmetry conv)
                if c > 1:
                    first[k].add_(c)
        _bn_touched.clear()


class _BnState:
    """Per-call batch statistics of one BatchNorm (scale/shift for the forward pre-op, mean/invstd for the backward).
    ``acc`` = accumulator block with the sums of act(x) / act(x)^2 when the producer of x already took them in its epilogue."""

    def __init__(self, x, relu, bn, L, acc=None):
        B, Cn = x.shape[0], x.shape[1]
        dev = x.device
        st = _stream(x)
        if acc is None:
            acc = _bn_acc(Cn, dev)
            _call("mural_op_bn_stats", x, B, Cn, L, int(relu), acc, st)
        self.scale, self.shift, self.mean, self.invstd = torch.empty((4, Cn), dtype=torch.float32, device=dev).unbind(0)
        _call("mural_op_bn_finalize", acc, float(B * L), Cn, _f32(bn.weight), _f32(bn.bias), EPS, MOMENTUM,
              bn.running_mean, bn.running_var, self.scale, self.shift, self.mean, self.invstd, st)
        _bn_tick(bn)


_part_cache = {}


def _wgrad_part(device):
    """Scratch for the per-workgroup weight-gradient partial rows (written and consumed inside one backward call)."""
    t = _part_cache.get(device)
    if t is None:
        t = _part_cache[device] = torch.empty(int(_lib.lib().mural_op_conv32_wgrad_scratch()), dtype=torch.float32, device=device)
    return t


class BnConv(torch.autograd.Function):
    """y = Conv1d(BN(act(x))) [+ ReLU] [+ res1 + res2], act = ReLU or identity, 32->32 channels, k=3, pad=1.

    ``stats_in``: batch sums of act(x) taken by the producer of x (skips the statistics pass); ``stats_out`` (None / False /
    True): also return the batch sums of y (True: of relu(y)) for the BatchNorm that consumes y, taken in the conv epilogue.
    Returns (y, sums or an empty tensor).  The 32-channel MFMA path is one C call per direction."""

    @staticmethod
    def forward(ctx, x, gamma, beta, weight, bias, res1, res2, bn, pre_relu, post_relu, stats_in=None, stats_out=None):
        x = x.contiguous()
        B, Cn, L = x.shape
        dev = x.device
        st = _stream(x)
        y = torch.empty((B, weight.shape[0], L), dtype=torch.float32, device=dev)
        mfma = tuple(weight.shape) == (32, 32, 3) and bool(_lib.lib().mural_op_conv32_supported(L))
        want = stats_out is not None
        acc_out = _bn_acc(weight.shape[0], dev) if want else torch.empty(0, dtype=torch.float32, device=dev)
        if mfma:      # fp32 MFMA implicit GEMM (csrc/conv32_mfma.hip)
            state = torch.empty((4, Cn), dtype=torch.float32, device=dev)
            acc = stats_in if stats_in is not None else _bn_acc(Cn, dev)
            _call("mural_op_bnconv32_fwd", x, B, L, int(pre_relu), acc, int(stats_in is not None), _f32(bn.weight), _f32(bn.bias),
                  EPS, MOMENTUM, bn.running_mean, bn.running_var, state, _f32(weight), _f32(bias), int(post_relu), _p(res1),
                  _p(res2), acc_out if want else None, int(bool(stats_out)), y, st)
            _bn_tick(bn)
        else:         # generic direct conv (csrc/conv1d.hip)
            bs = _BnState(x, pre_relu, bn, L, stats_in)
            state = torch.stack([bs.scale, bs.shift, bs.mean, bs.invstd])
            wt = torch.empty_like(weight)
            _call("mural_op_relayout", _f32(weight), wt, weight.shape[0], weight.shape[1], weight.shape[2], 0, st)
            _call("mural_op_conv1d", x, wt, _f32(bias), y, B, Cn, weight.shape[0], L, weight.shape[2], state[0], state[1],
                  int(pre_relu), int(post_relu), _p(res1), _p(res2), st)
            if want:
                _call("mural_op_bn_stats", y, B, weight.shape[0], L, int(bool(stats_out)), acc_out, st)
        ctx.save_for_backward(x, gamma, weight, y if post_relu else None, state)
        ctx.flags = (pre_relu, post_relu, res1 is not None, res2 is not None, mfma)
        ctx.mark_non_differentiable(acc_out)
        return y, acc_out

    @staticmethod
    def backward(ctx, dy, _dacc):
        x, gamma, weight, y, state = ctx.saved_tensors
        pre_relu, post_relu, has_r1, has_r2, mfma = ctx.flags
        dy = dy.contiguous()
        B, Cn, L = x.shape
        dev = x.device
        st = _stream(x)
        dres = dy
        if post_relu:
            g = torch.empty_like(dy)
            _call("mural_op_relu_mask", dy, y, dy.numel(), g, st)
            dy = g
        dW = torch.empty_like(weight)
        small = torch.empty((3, Cn), dtype=torch.float32, device=dev)          # db | dgamma | dbeta
        db, dgamma, dbeta = small[0], small[1], small[2]
        dz = torch.empty_like(x)
        dx = torch.empty_like(x)
        acc = _bn_acc(Cn, dev)
        if mfma:      # one pass over dy: weight / bias gradient, input gradient, BatchNorm-backward sums; then the BN backward
            part = _wgrad_part(dev)
            _call("mural_op_bnconv32_bwd", dy, x, B, L, int(pre_relu), state, _f32(gamma), _f32(weight), acc, part, part.numel(),
                  dz, None, None, dW, db, dx, dgamma, dbeta, st)
        else:
            scale, shift, mean, invstd = state[0], state[1], state[2], state[3]
            wt = torch.empty_like(weight)
            part = torch.empty(1024 * (weight.numel() + weight.shape[0]), dtype=torch.float32, device=dev)
            _call("mural_op_conv_wgrad", dy, x, B, Cn, L, weight.shape[2], scale, shift, int(pre_relu), dW, db, part,
                  part.numel(), st)
            _call("mural_op_relayout", _f32(weight), wt, weight.shape[0], weight.shape[1], weight.shape[2], 1, st)
            _call("mural_op_conv1d", dy, wt, None, dz, B, weight.shape[0], Cn, L, weight.shape[2], None, None, 0, 0, None, None,
                  st)
            _call("mural_op_bn_backward", dz, x, B, Cn, L, int(pre_relu), mean, invstd, _f32(gamma), acc, 0, None, None, dx,
                  dgamma, dbeta, st)
        return (dx, dgamma, dbeta, dW, db, (dres if has_r1 else None), (dres if has_r2 else None), None, None, None, None,
                None)


class Bn2d(torch.autograd.Function):
    """y = BN(act(x)) on (B, C) features (batch statistics over B)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, bn, pre_relu):
        x = x.contiguous()
        B, Cn = x.shape
        state = _BnState(x, pre_relu, bn, 1)
        y = torch.empty_like(x)
        _call("mural_op_bn_apply", x, B, Cn, 1, int(pre_relu), state.scale, state.shift, y, _stream(x))
        ctx.save_for_backward(x, gamma, state.mean, state.invstd)
        ctx.pre_relu = pre_relu
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, mean, invstd = ctx.saved_tensors
        B, Cn = x.shape
        acc = _bn_acc(Cn, x.device)
        dx = torch.empty_like(x)
        dgamma = torch.empty(Cn, dtype=torch.float32, device=x.device)
        dbeta = torch.empty(Cn, dtype=torch.float32, device=x.device)
        _call("mural_op_bn_backward", dy.contiguous(), x, B, Cn, 1, int(ctx.pre_relu), mean, invstd, _f32(gamma), acc, 0, None,
              None, dx, dgamma, dbeta, _stream(x))
        return dx, dgamma, dbeta, None, None


class MaxPool(torch.autograd.Function):
    """MaxPool1d(k, s, p) (floor mode, -inf padding); k = None -> global max over L (returns (B, C))."""

    @staticmethod
    def forward(ctx, x, k, s, p):
        x = x.contiguous()
        B, Cn, L = x.shape
        glob = k is None
        if glob:
            k, s, p = L, L, 0          # one window per row: the backward is a gather (no zero fill, no atomics)
        Lout = (L + 2 * p - k) // s + 1
        y = torch.empty((B, Cn, Lout), dtype=torch.float32, device=x.device)
        arg = torch.empty((B, Cn, Lout), dtype=torch.int32, device=x.device)
        _call("mural_op_maxpool_fwd", x, B * Cn, L, k, s, p, y, arg, _stream(x))
        ctx.save_for_backward(arg)
        ctx.dims = (B, Cn, L, Lout, glob, k, s, p)
        return y.reshape(B, Cn) if glob else y

    @staticmethod
    def backward(ctx, dy):
        (arg,) = ctx.saved_tensors
        B, Cn, L, Lout, glob, k, s, p = ctx.dims
        # disjoint windows (stride >= kernel, every pool of the model): a gather writes all of dx, no zero fill / atomics
        dx = (torch.zeros if s < k else torch.empty)((B, Cn, L), device=dy.device)
        _call("mural_op_maxpool_bwd", dy.contiguous(), arg, B * Cn, L, Lout, k, s, p, dx, _stream(dy))
        return dx, None, None, None


def _first_plan(Cn, pk):
    """(tab floats, arg bytes per pooled output, backward scratch floats) of the first-layer kernels for this shape."""
    import ctypes as C
    t, a, s = C.c_int64(0), C.c_int64(0), C.c_int64(0)
    _lib.check(_lib.lib().mural_op_first_plan(Cn, pk, C.byref(t), C.byref(a), C.byref(s)))
    return t.value, a.value, s.value


class FirstLayerPool(torch.autograd.Function):
    """maxpool1(Conv1d(BN(one-hot))) of one tower from window symbols: 3-mer / per-tap symbol tables rebuilt from the batch
    statistics of this step, arg-max saved (one byte per pooled output on the table path)."""

    @staticmethod
    def forward(ctx, sym, gamma, beta, weight, bias, bn, col0, L1, pool):
        B, Lwin = sym.shape
        Cn = weight.shape[0]
        pk, ps, pp = pool
        L2 = (L1 + 2 * pp - pk) // ps + 1
        dev = sym.device
        tab_floats, arg_bytes, _ = _first_plan(Cn, pk)
        counts = torch.zeros(16, dtype=torch.int64, device=dev)
        tab = torch.empty(tab_floats, dtype=torch.float32, device=dev)
        y = torch.empty((B, Cn, L2), dtype=torch.float32, device=dev)
        arg = torch.empty(B * Cn * L2 * arg_bytes, dtype=torch.uint8, device=dev)
        _call("mural_op_first_fwd", sym, B, Lwin, col0, L1, Cn, pk, ps, pp, _f32(gamma), _f32(beta), _f32(weight), _f32(bias),
              EPS, MOMENTUM, bn.running_mean, bn.running_var, counts, tab, y, arg, _stream(sym))
        _bn_tick(bn)
        ctx.save_for_backward(sym, arg, tab, weight)
        ctx.dims = (col0, L1, pool)
        return y

    @staticmethod
    def backward(ctx, dy):
        sym, arg, tab, weight = ctx.saved_tensors
        col0, L1, (pk, ps, pp) = ctx.dims
        B, Lwin = sym.shape
        Cn = weight.shape[0]
        dev = sym.device
        scratch = torch.empty(_first_plan(Cn, pk)[2], dtype=torch.float32, device=dev)
        dW = torch.empty_like(weight)
        db = torch.empty(Cn, dtype=torch.float32, device=dev)
        dgamma = torch.empty(4, dtype=torch.float32, device=dev)
        dbeta = torch.empty(4, dtype=torch.float32, device=dev)
        _call("mural_op_first_bwd", dy.contiguous(), arg, sym, B, Lwin, col0, L1, Cn, pk, ps, pp, tab, _f32(weight), scratch, dW,
              db, dgamma, dbeta, _stream(sym))
        return None, dgamma, dbeta, dW, db, None, None, None, None


class Linear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        x = x.contiguous()
        B, I = x.shape
        O = weight.shape[0]
        y = torch.empty((B, O), dtype=torch.float32, device=x.device)
        _call("mural_op_linear_fwd", x, _f32(weight), _f32(bias), B, I, O, y, _stream(x))
        ctx.save_for_backward(x, weight)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        B, I = x.shape
        O = weight.shape[0]
        dx = torch.empty_like(x)
        dW = torch.empty_like(weight)
        db = torch.empty(O, dtype=torch.float32, device=x.device)
        _call("mural_op_linear_bwd", dy.contiguous(), x, _f32(weight), B, I, O, dx, dW, db, _stream(x))
        return dx, dW, db


class Embedding(torch.autograd.Function):
    """concat_i emb[cat[:, i]] with ONE shared (rows, 5) table (model_snv.py:322, :452-454)."""

    @staticmethod
    def forward(ctx, cat, table):
        cat = cat.contiguous()
        B, cols = cat.shape
        y = torch.empty((B, cols * 5), dtype=torch.float32, device=table.device)
        _call("mural_op_embedding_fwd", cat, _f32(table), B, cols, table.shape[0], y, _stream(table))
        ctx.save_for_backward(cat)
        ctx.rows = table.shape[0]
        return y

    @staticmethod
    def backward(ctx, dy):
        (cat,) = ctx.saved_tensors
        B, cols = cat.shape
        dE = torch.zeros((ctx.rows, 5), dtype=torch.float32, device=dy.device)
        _call("mural_op_embedding_bwd", cat, dy.contiguous(), B, cols, ctx.rows, dE, _stream(dy))
        return None, dE


class Dropout(torch.autograd.Function):
    """Inverted dropout with a counter-based generator; the mask is regenerated from the seed in the backward."""

    @staticmethod
    def forward(ctx, x, p, seed, seed_dev=None):
        x = x.contiguous()
        y = torch.empty_like(x)
        _call("mural_op_dropout", x, x.numel(), float(p), C.c_uint64(seed), seed_dev, y, _stream(x))
        ctx.p, ctx.seed, ctx.seed_dev = float(p), seed, seed_dev
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        dx = torch.empty_like(dy)
        _call("mural_op_dropout", dy, dy.numel(), ctx.p, C.c_uint64(ctx.seed), ctx.seed_dev, dx, _stream(dy))
        return dx, None, None, None


_device_seed = None


def set_device_seed(t):
    """A device-resident uint64 step counter (int64 tensor of one element) added to every dropout seed, or None.  A captured
    training step (mural_amd.train.GraphedTrainStep) bakes the host-drawn seeds into the graph and advances this counter
    inside it, so every replay draws fresh masks."""
    global _device_seed
    _device_seed = t


def dropout(x, p, training=True):
    if not training or p <= 0.0:
        return x
    seed = int(torch.randint(0, 2 ** 62, (1,)).item())     # drawn from torch's CPU generator: torch.manual_seed applies
    return Dropout.apply(x, p, seed, _device_seed)


class Head(torch.autograd.Function):
    """log(clamp((softmax(local) + (softmax(mid) + softmax(large)) / 2) / 2, 1e-9)); local=None for Network1."""

    @staticmethod
    def forward(ctx, loc, mid, lar):
        B, nc = mid.shape
        out = torch.empty((B, nc), dtype=torch.float32, device=mid.device)
        loc_c = None if loc is None else loc.contiguous()
        mid, lar = mid.contiguous(), lar.contiguous()
        _call("mural_op_head_fwd", loc_c, mid, lar, B, nc, out, _stream(mid))
        ctx.save_for_backward(mid, lar, *([] if loc_c is None else [loc_c]))
        return out

    @staticmethod
    def backward(ctx, dout):
        saved = ctx.saved_tensors
        mid, lar = saved[0], saved[1]
        loc = saved[2] if len(saved) > 2 else None
        B, nc = mid.shape
        dmid, dlar = torch.empty_like(mid), torch.empty_like(lar)
        dloc = None if loc is None else torch.empty_like(loc)
        _call("mural_op_head_bwd", loc, mid, lar, dout.contiguous(), B, nc, dloc, dmid, dlar, _stream(mid))
        return dloc, dmid, dlar


_pending_checks = []
_status_host = {}
captured_status = []


def dense_to_symbols(distal_x):
    """(B, 4, L) MuRaL one-hot / IUPAC-fraction tensor -> (B, L) uint8 symbols.  Anything else raises ValueError from
    ``flush_input_checks`` (called by the model forward after its launches are enqueued: the host then waits for this early
    kernel only instead of draining the device before the forward starts)."""
    x = distal_x.to(torch.float32).contiguous()
    B, _, L = x.shape
    dev = x.device
    sym = torch.empty((B, L), dtype=torch.uint8, device=dev)
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    _call("mural_op_dense_to_symbols", x, B, L, sym, status, _stream(x))
    if torch.cuda.is_current_stream_capturing():
        captured_status.append(status)          # read by the owner of the graph after a replay (GraphedTrainStep)
        return sym
    host = _status_host.get(dev)
    if host is None:
        host = _status_host[dev] = torch.zeros(1, dtype=torch.int32).pin_memory()
    host.copy_(status, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(dev))
    _pending_checks.append((ev, host))
    return sym


def flush_input_checks():
    if os.environ.get("MURAL_DEBUG_NO_INPUT_CHECK"):      # diagnostic (tools/host_vs_gpu_train.py): never wait for the device
        _pending_checks.clear()
    if _pending_checks and torch.cuda.is_current_stream_capturing():
        return                                            # a captured step: the owner of the graph checks after the replay
    while _pending_checks:
        ev, host = _pending_checks.pop()
        ev.synchronize()
        if int(host[0]) != 0:
            raise ValueError("distal_input holds a column that is not a MuRaL one-hot / IUPAC-fraction encoding")


# ---------------------------------------------------------------------------------------------------------------
# model composition (training mode)
# ---------------------------------------------------------------------------------------------------------------
def tower_forward(mod, sfx, sym, col0, L1, pools, dropout_p, training=True):
    g = lambda n: getattr(mod, n + sfx)
    bn_in, conv_in = g("conv1")[0], g("conv1")[1]
    x0 = FirstLayerPool.apply(sym, bn_in.weight, bn_in.bias, conv_in.weight, conv_in.bias, bn_in, col0, L1, pools[0])

    def bnconv(x, bn, conv, res1=None, res2=None, pre_relu=True, post_relu=False, stats_in=None, stats_out=None):
        y, acc = BnConv.apply(x, bn.weight, bn.bias, conv.weight, conv.bias, res1, res2, bn, pre_relu, post_relu, stats_in,
                              stats_out)
        return y, (acc if stats_out is not None else None)

    def res_blocks(rbs, x_in, st_in):
        # every conv takes the batch sums of relu(its output) in its epilogue for the BatchNorm of the next layer
        rb0, rb1 = rbs[0], rbs[1]
        h, st = bnconv(x_in, rb0.bn1, rb0.conv1, stats_in=st_in, stats_out=True)
        x1, st = bnconv(h, rb0.bn2, rb0.conv2, res1=x_in, stats_in=st, stats_out=True)
        h, st = bnconv(x1, rb1.bn1, rb1.conv1, stats_in=st, stats_out=True)
        # second block's own residual (x1) plus the outer skip (x_in), model_snv.py:477-479
        return bnconv(h, rb1.bn2, rb1.conv2, res1=x1, res2=x_in, stats_in=st)[0]

    y = res_blocks(g("RBs1"), x0, None)
    p2 = MaxPool.apply(y, *pools[1])
    x0b, st = bnconv(p2, g("conv2")[0], g("conv2")[1], pre_relu=False, stats_out=True)
    y = res_blocks(g("RBs2"), x0b, st)
    p3 = MaxPool.apply(y, *pools[2])
    c3 = bnconv(p3, g("conv3")[0], g("conv3")[1], pre_relu=False, post_relu=True)[0]
    feat = MaxPool.apply(c3, None, None, None)
    fc = mod.distal_fc1 if sfx == "" else mod.distal_fc2
    f = Bn2d.apply(feat, fc[0].weight, fc[0].bias, fc[0], False)
    f = dropout(f, dropout_p, training)
    return Linear.apply(f, fc[2].weight, fc[2].bias)


def local_forward(mod, cat, out_layer, emb_p, lin_ps, training=True):
    h = Embedding.apply(cat, mod.emb_layer.weight)
    h = dropout(h, emb_p, training)
    for lin, bn, p in zip(mod.lin_layers, mod.bn_layers, lin_ps):
        h = Linear.apply(h, lin.weight, lin.bias)
        h = Bn2d.apply(h, bn.weight, bn.bias, bn, True)          # order Linear -> ReLU -> BN (model_snv.py:466-467)
        h = dropout(h, p, training)
    return Linear.apply(h, out_layer.weight, out_layer.bias)
