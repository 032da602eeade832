"""Training-mode forward of ``UNet_Small`` on the HIP building blocks (``mural_op_convg_*``, ``mural_op_act_*`` of
csrc/indel_train.hip; BatchNorm / Linear / Dropout / global max of csrc/train_ops.hip).

The reference runs stock autograd over ``torch.nn`` modules (MuRaL/model/model_indel.py:151-176 under ``model.train()``,
MuRaL/training.py:404-450).  Here each layer is a ``torch.autograd.Function`` whose forward and backward launch the
hand-written kernels; torch supplies the tensors, the stream, the autograd graph and the residual adds / flips.
BatchNorm uses batch statistics and updates ``running_mean`` / ``running_var`` / ``num_batches_tracked`` exactly like
``nn.BatchNorm1d`` (the strand-symmetrising ``conv`` is applied twice per forward, so its BatchNorm is updated twice, as in
the reference).
"""
import torch

from .. import _lib
from . import train_ops as T

ACT_RELU, ACT_SILU, ACT_SOFTPLUS = 1, 2, 3

_scratch = {}


def _bwd_scratch(device, n):
    t = _scratch.get(device)
    if t is None or t.numel() < n:
        t = _scratch[device] = torch.empty(n, dtype=torch.float32, device=device)
    return t


class Conv(torch.autograd.Function):
    """y = Conv1d(Upsample(scale_factor=up)(x)) with torch-layout weight (Cout, Cin, K), optional bias, stride, zero padding."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, pad, up):
        x = x.contiguous()
        B, Cin, Lin = x.shape
        Cout, _, K = weight.shape
        Lout = int(_lib.lib().mural_op_convg_out_length(Lin, K, stride, pad, up))
        if Lout < 1:
            raise ValueError(f"Conv1d: kernel {K} / padding {pad} do not fit an input of length {Lin * up}")
        y = torch.empty((B, Cout, Lout), dtype=torch.float32, device=x.device)
        wt = torch.empty(weight.numel(), dtype=torch.float32, device=x.device)
        T._call("mural_op_convg_fwd", x, T._f32(weight), None if bias is None else T._f32(bias), wt, y, B, Cin, Lin, Cout, K, stride,
                pad, up, T._stream(x))
        ctx.save_for_backward(x, weight)
        ctx.geom = (stride, pad, up, bias is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        stride, pad, up, has_bias = ctx.geom
        dy = dy.contiguous()
        B, Cin, Lin = x.shape
        Cout, _, K = weight.shape
        dev = x.device
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dW = torch.empty_like(weight)
        db = torch.empty(Cout, dtype=torch.float32, device=dev) if has_bias else None
        part = _bwd_scratch(dev, int(_lib.lib().mural_op_convg_bwd_scratch(Cin, Cout, K)))
        T._call("mural_op_convg_bwd", dy, x, T._f32(weight), B, Cin, Lin, Cout, K, stride, pad, up, dx, dW, db, part, part.numel(),
                T._stream(x))
        return dx, dW, db, None, None, None


class BatchNorm(torch.autograd.Function):
    """y = BatchNorm1d(act(x)) on (B, C, L) with batch statistics (running statistics updated in place); act = ReLU with
    ``pre_relu`` (the SNV towers' ReLU -> BN order), identity otherwise."""

    @staticmethod
    def forward(ctx, x, gamma, beta, bn, pre_relu=False):
        x = x.contiguous()
        B, Cn, L = x.shape
        state = T._BnState(x, bool(pre_relu), bn, L)
        y = torch.empty_like(x)
        T._call("mural_op_bn_apply", x, B, Cn, L, int(pre_relu), state.scale, state.shift, y, T._stream(x))
        ctx.save_for_backward(x, gamma, state.mean, state.invstd)
        ctx.pre_relu = bool(pre_relu)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, mean, invstd = ctx.saved_tensors
        B, Cn, L = x.shape
        acc = T._bn_acc(Cn, x.device)
        dx = torch.empty_like(x)
        dgamma = torch.empty(Cn, dtype=torch.float32, device=x.device)
        dbeta = torch.empty(Cn, dtype=torch.float32, device=x.device)
        T._call("mural_op_bn_backward", dy.contiguous(), x, B, Cn, L, int(ctx.pre_relu), mean, invstd, T._f32(gamma), acc, 0, None,
                None, dx, dgamma, dbeta, T._stream(x))
        return dx, dgamma, dbeta, None, None


class Act(torch.autograd.Function):
    """ReLU / SiLU / Softplus; the backward re-derives the slope from the saved input."""

    @staticmethod
    def forward(ctx, x, kind):
        x = x.contiguous()
        y = torch.empty_like(x)
        T._call("mural_op_act_fwd", x, x.numel(), kind, y, T._stream(x))
        ctx.save_for_backward(x)
        ctx.kind = kind
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dx = torch.empty_like(x)
        T._call("mural_op_act_bwd", dy.contiguous(), x, x.numel(), ctx.kind, dx, T._stream(x))
        return dx, None


class ConvBn(torch.autograd.Function):
    """z = act(BatchNorm1d(Conv1d(Upsample(up)(x)))) [+ res1] [+ res2] in training mode: the unit of the U-Net (model_indel.py:6-19,
    :117-123) as ONE autograd node and one library call per direction (``mural_op_convg_bn_fwd`` / ``_bwd``).  The BatchNorm output
    is not stored: the backward re-derives the activation's slope from the saved conv output."""

    @staticmethod
    def forward(ctx, x, weight, bias, gamma, beta, bn, stride, pad, up, act, res1, res2, wt_fwd=None, wt_dgrad=None):
        x = x.contiguous()
        B, Cin, Lin = x.shape
        Cout, _, K = weight.shape
        dev = x.device
        Lout = int(_lib.lib().mural_op_convg_out_length(Lin, K, stride, pad, up))
        if Lout < 1:
            raise ValueError(f"Conv1d: kernel {K} / padding {pad} do not fit an input of length {Lin * up}")
        y0 = torch.empty((B, Cout, Lout), dtype=torch.float32, device=dev)
        z = torch.empty_like(y0)
        # wt_fwd / wt_dgrad: this step's layouts of `weight` from the model-wide relayout launch (_WeightLayouts); without them the
        # calls derive the layouts themselves
        wt = wt_fwd if wt_fwd is not None else torch.empty(weight.numel(), dtype=torch.float32, device=dev)
        state = torch.empty((4, Cout), dtype=torch.float32, device=dev)
        for r in (res1, res2):
            if r is not None and (r.shape != y0.shape or not r.is_contiguous() or r.dtype is not torch.float32):
                raise ValueError("ConvBn: a residual must be a contiguous float32 tensor of the output's shape")
        T._call("mural_op_convg_bn_fwd", x, None if wt_fwd is not None else T._f32(weight), None if bias is None else T._f32(bias), wt, y0,
                B, Cin, Lin, Cout, K, stride,
                pad, up, T._f32(gamma), T._f32(beta), T.EPS, T.MOMENTUM, bn.running_mean, bn.running_var, T._bn_acc(Cout, dev), state,
                act, res1, res2, z, T._stream(x))
        T._bn_tick(bn)
        ctx.save_for_backward(x, weight, y0, state, gamma)
        ctx.geom = (stride, pad, up, act, bias is not None, res1 is not None, res2 is not None)
        ctx.wt_dgrad = wt_dgrad        # not a saved tensor: a scratch view that the next step's relayout launch rewrites
        return z

    @staticmethod
    def backward(ctx, dz):
        x, weight, y0, state, gamma = ctx.saved_tensors
        stride, pad, up, act, has_bias, has_r1, has_r2 = ctx.geom
        dz = dz.contiguous()
        B, Cin, Lin = x.shape
        Cout, _, K = weight.shape
        dev = x.device
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dW = torch.empty_like(weight)
        db = torch.empty(Cout, dtype=torch.float32, device=dev) if has_bias else None
        dgb = torch.empty((2, Cout), dtype=torch.float32, device=dev)
        dy0 = torch.empty_like(y0)
        part = _bwd_scratch(dev, int(_lib.lib().mural_op_convg_bwd_scratch(Cin, Cout, K)))
        T._call("mural_op_convg_bn_bwd", dz, x, T._f32(weight), y0, state, T._f32(gamma), B, Cin, Lin, Cout, K, stride, pad, up, act,
                T._bn_acc(Cout, dev), dy0, dx, dW, db, dgb[0], dgb[1], part, part.numel(), ctx.wt_dgrad, T._stream(x))
        return dx, dW, db, dgb[0], dgb[1], None, None, None, None, None, dz if has_r1 else None, dz if has_r2 else None, None, None


class _WeightLayouts:
    """This step's kernel layouts of every Conv1d weight of a model -- forward [Cin][K][Cout] and, for stride-1 layers, the input
    gradient's [Cout][K flipped][Cin] -- written by ONE launch per step (``mural_op_relayout_multi``) instead of one or two per
    layer.  The job table lives on the device and is rebuilt when a weight's storage moves."""

    def __init__(self, model):
        self.convs = [m for m in model.modules() if isinstance(m, torch.nn.Conv1d)]
        self.key = None

    def _build(self, dev):
        total = sum(c.weight.numel() for c in self.convs)
        n_dg = sum(c.weight.numel() for c in self.convs if int(c.stride[0]) == 1)
        self.buf = torch.empty(total + n_dg, dtype=torch.float32, device=dev)
        jobs = (_lib.MuralRelayoutJob * len(self.convs))()
        self.views = {}
        off_f, off_d, start = 0, total, 0
        for j, c in enumerate(self.convs):
            w = c.weight
            n = w.numel()
            if w.dtype is not torch.float32 or not w.is_contiguous():
                raise RuntimeError("the HIP training step needs contiguous float32 conv weights")
            fwd = self.buf[off_f:off_f + n]
            dg = self.buf[off_d:off_d + n] if int(c.stride[0]) == 1 else None
            jobs[j].W, jobs[j].wt_fwd, jobs[j].wt_dgrad = w.data_ptr(), fwd.data_ptr(), (dg.data_ptr() if dg is not None else None)
            jobs[j].Cout, jobs[j].Cin, jobs[j].K, jobs[j].start = w.shape[0], w.shape[1], w.shape[2], start
            self.views[id(c)] = (fwd, dg)
            off_f += n
            off_d += n if dg is not None else 0
            start += n
        raw = torch.frombuffer(bytearray(bytes(jobs)), dtype=torch.uint8)
        self.jobs_dev = raw.to(dev)
        self.n, self.total = len(self.convs), total

    def refresh(self, dev, stream):
        key = (dev, tuple(c.weight.data_ptr() for c in self.convs))
        if key != self.key:
            self._build(dev)
            self.key = key
        T._call("mural_op_relayout_multi", self.jobs_dev, self.n, self.total, stream)
        return self.views


def _layouts(model, x):
    lay = getattr(model, "_train_weight_layouts", None)
    if lay is None:
        lay = model._train_weight_layouts = _WeightLayouts(model)
    return lay.refresh(x.device, T._stream(x))


def _cba(x, conv, bn, act=0, up=1, res1=None, res2=None, wl=None):
    fwd, dg = wl[id(conv)] if wl is not None else (None, None)
    return ConvBn.apply(x, conv.weight, conv.bias, bn.weight, bn.bias, bn, int(conv.stride[0]), int(conv.padding[0]), int(up), act,
                        res1, res2, fwd, dg if int(up) == 1 else None)


def _conv(x, conv, up=1):
    return Conv.apply(x, conv.weight, conv.bias, int(conv.stride[0]), int(conv.padding[0]), int(up))


def _bn(x, bn):
    return BatchNorm.apply(x, bn.weight, bn.bias, bn, False)


def _conv_block(x, cb, skip=None, wl=None):
    """[skip +] x + BN(Conv1x1(SiLU(BN(Conv5(x)))))   (model_indel.py:6-19; the decoder adds the encoder's skip tensor, :168-170)"""
    seq = cb.conv
    h = _cba(x, seq[0], seq[1], ACT_SILU, wl=wl)
    return _cba(h, seq[3], seq[4], 0, res1=x, res2=skip, wl=wl)


def unet_forward_train(model, x):
    """``UNet_Small.forward`` (model_indel.py:151-176) in training mode."""
    try:
        return _unet_forward_train(model, x)
    except BaseException:
        T._bn_touched.clear()          # a failed forward must not leave its BatchNorm counter ticks to the next one
        raise


def _unet_forward_train(model, x):
    out = x
    wl = _layouts(model, x)
    if model.use_reverse:
        sym = lambda t: _cba(t, model.conv[0], model.conv[1], wl=wl)         # noqa: E731
        out = sym(out) + sym(out.flip([1, 2])).flip([2])
    encodings = []
    for lconv, conv in zip(model.uplblocks, model.upblocks):
        out = _conv_block(_cba(out, lconv[0], lconv[1], wl=wl), conv[0], wl=wl)
        encodings.append(out)
    for enc, lconv, conv in zip(reversed(encodings[:-1]), model.downlblocks, model.downblocks):
        up = int(lconv[0].scale_factor)
        out = _conv_block(_cba(out, lconv[1], lconv[2], up=up, wl=wl), conv[0], skip=enc, wl=wl)   # = enc + (x + BN(...)): same sum, same order
    oc = model.out_conv
    out = _cba(out, oc[0], oc[1], ACT_RELU, wl=wl)
    out = Act.apply(_conv(out, oc[3]), ACT_SOFTPLUS)
    feat = T.MaxPool.apply(out, None, None, None)
    fc = model.out_fc
    f = T.Bn2d.apply(feat, fc[0].weight, fc[0].bias, fc[0], False)
    f = T.dropout(f, float(fc[1].p), True)
    res = Act.apply(T.Linear.apply(f, fc[2].weight, fc[2].bias), ACT_SOFTPLUS)
    T.flush_bn_ticks()
    return res


# ------------------------------------------------------------------------------------------------------------------
# SNV towers of any width / kernel size (the MFMA training kernels of train_ops.py serve the shipped 32-channel, k=3 shape)
# ------------------------------------------------------------------------------------------------------------------
def snv_tower_forward_train(mod, sfx, x, pools, dropout_p):
    """One conv tower of Network1/2 (model_snv.py:473-493 / :496-513) in training mode from the dense (B, 4, L) window, on the
    general per-layer ops.  Same layer order as ``train_ops.tower_forward``."""
    g = lambda n: getattr(mod, n + sfx)          # noqa: E731

    def bnconv(t, seq, pre_relu=False):
        return _conv(BatchNorm.apply(t, seq[0].weight, seq[0].bias, seq[0], pre_relu), seq[1])

    def res_blocks(rbs, t):
        out = t
        for rb in rbs:
            h = _conv(BatchNorm.apply(out, rb.bn1.weight, rb.bn1.bias, rb.bn1, True), rb.conv1)
            h = _conv(BatchNorm.apply(h, rb.bn2.weight, rb.bn2.bias, rb.bn2, True), rb.conv2)
            out = out + h
        return out + t

    out = T.MaxPool.apply(bnconv(x.contiguous(), g("conv1")), *pools[0])
    out = T.MaxPool.apply(res_blocks(g("RBs1"), out), *pools[1])
    out = bnconv(out, g("conv2"))
    out = T.MaxPool.apply(res_blocks(g("RBs2"), out), *pools[2])
    out = Act.apply(bnconv(out, g("conv3")), ACT_RELU)
    feat = T.MaxPool.apply(out, None, None, None)
    fc = mod.distal_fc1 if sfx == "" else mod.distal_fc2
    f = T.Bn2d.apply(feat, fc[0].weight, fc[0].bias, fc[0], False)
    f = T.dropout(f, dropout_p, True)
    return T.Linear.apply(f, fc[2].weight, fc[2].bias)
