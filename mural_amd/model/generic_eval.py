"""Eval-mode forward of the SNV networks for shapes the fused tower kernel is not built for (``CNN_out_channels`` != 32,
``CNN_kernel_size`` != 3, very long windows): the same layer sequence as MuRaL/model/model_snv.py:449-523 (Network2;
:226-287 Network1), one HIP launch per layer (``mural_op_bn_apply`` with the running statistics as a per-channel affine,
``mural_op_convg_fwd``, ``mural_op_maxpool_fwd``, ``mural_op_linear_fwd``, ``mural_op_embedding_fwd``, ``mural_op_head_fwd``).
It exists for completeness of the drop-in (any ``model_choice`` configuration evaluates); every shipped checkpoint and the
benchmark configuration take the fused path in ``model_snv.py``.  No CPU path here either.
"""
import weakref

import torch

from . import train_ops as T

# Per-module caches of what an eval-mode forward derives from the parameters: the BatchNorm affines and the conv weights in the
# kernels' [Cin][K][Cout] layout.  Keyed by the module, validated by the tensors' version counters (an optimiser step, a
# load_state_dict or a running-statistics update bumps them) and their storage: recomputing them per call was 5 tiny launches per
# BatchNorm and a relayout launch per conv -- most of the launches of this path.
_affine_cache = weakref.WeakKeyDictionary()
_wt_cache = weakref.WeakKeyDictionary()


def _stamp(*ts):
    return tuple((t.data_ptr(), t._version, t.device) for t in ts)


def invalidate(model=None):
    """Drop the cached affines / weight layouts of `model`'s modules (all caches when None).  The version stamps above see every
    in-place update made through the tensor itself; an edit through ``.data`` (``bn.weight.data.mul_()``, ``conv.weight.data.copy_()``:
    older init / EMA code) bumps neither the stamp nor the pointer -- the model's ``invalidate_folded()`` (called by ``train()``
    transitions, ``load_state_dict`` and ``_apply``; public for hand edits) comes through here."""
    if model is None:
        _affine_cache.clear()
        _wt_cache.clear()
        return
    for m in model.modules():
        _affine_cache.pop(m, None)
        _wt_cache.pop(m, None)


class _NoBn:
    """Stands for a BatchNorm that was applied already (scale 1, shift 0)."""
    _made = {}

    def __init__(self, channels, device):
        key = (int(channels), str(device))
        if key not in self._made:
            self._made[key] = (torch.ones(channels, device=device), torch.zeros(channels, device=device))
        self.pair = self._made[key]


def _affine(bn):
    """eval-mode BatchNorm as y = scale * x + shift"""
    if isinstance(bn, _NoBn):
        return bn.pair
    key = _stamp(bn.weight, bn.bias, bn.running_mean, bn.running_var) + (float(bn.eps),)
    hit = _affine_cache.get(bn)
    if hit is not None and hit[0] == key:
        return hit[1], hit[2]
    scale = (bn.weight.detach() / torch.sqrt(bn.running_var + bn.eps)).to(torch.float32).contiguous()
    shift = (bn.bias.detach() - bn.running_mean * scale).to(torch.float32).contiguous()
    _affine_cache[bn] = (key, scale, shift)
    return scale, shift


def _weights(conv):
    """(wt, bias): the conv weight re-laid-out as [Cin][K][Cout] (mural_op_relayout), cached per module"""
    key = _stamp(conv.weight) + (() if conv.bias is None else _stamp(conv.bias))
    hit = _wt_cache.get(conv)
    if hit is not None and hit[0] == key:
        return hit[1], hit[2]
    w = T._f32(conv.weight.detach())
    Cout, Cin, K = w.shape
    wt = torch.empty(w.numel(), device=w.device)
    T._call("mural_op_relayout", w, wt, Cout, Cin, K, 0, T._stream(w))
    bias = None if conv.bias is None else T._f32(conv.bias.detach()).clone()
    _wt_cache[conv] = (key, wt, bias)
    return wt, bias


def _bn(x, bn, relu):
    """BN(relu(x)) if relu else BN(x); x is (B, C, L) or (B, C)"""
    x = x.contiguous()
    B, Cn = x.shape[0], x.shape[1]
    L = x.shape[2] if x.dim() == 3 else 1
    scale, shift = _affine(bn)
    y = torch.empty_like(x)
    T._call("mural_op_bn_apply", x, B, Cn, L, int(relu), scale, shift, y, T._stream(x))
    return y


def _conv(x, conv):
    x = x.contiguous()
    B, Cin, L = x.shape
    w = T._f32(conv.weight.detach())
    Cout, _, K = w.shape
    pad = int(conv.padding[0])
    Lout = L + 2 * pad - K + 1
    if Lout < 1:
        raise ValueError(f"Conv1d: kernel {K} does not fit an input of length {L}")
    y = torch.empty((B, Cout, Lout), device=x.device)
    wt = torch.empty(w.numel(), device=x.device)
    T._call("mural_op_convg_fwd", x, w, None if conv.bias is None else T._f32(conv.bias.detach()), wt, y, B, Cin, L, Cout, K, 1, pad, 1,
            T._stream(x))
    return y


def _bn_conv(x, bn, pre_relu, conv, post_relu=False, res1=None, res2=None):
    """conv(BN(relu?(x))) [ReLU] [+ res1] [+ res2] in ONE launch when the conv is a stride-1 'same' conv (mural_op_conv1d applies the
    BatchNorm affine -- after the optional ReLU, before the zero padding, like nn.Conv1d behind a BatchNorm1d -- while it stages its
    input tile, and the residuals in its epilogue); the separate launches otherwise."""
    Cout, Cin, K = conv.weight.shape
    same = int(conv.stride[0]) == 1 and int(conv.dilation[0]) == 1 and int(conv.padding[0]) == (K - 1) // 2 and K % 2 == 1
    if not (same and Cout % 4 == 0 and x.dim() == 3):
        y = _conv(_bn(x, bn, pre_relu), conv)
        if post_relu:
            y = _relu(y)
        for r in (res1, res2):
            if r is not None:
                y = y + r
        return y
    x = x.contiguous()
    B, _, L = x.shape
    scale, shift = _affine(bn)
    wt, bias = _weights(conv)
    y = torch.empty((B, Cout, L), device=x.device)
    T._call("mural_op_conv1d", x, wt, bias, y, B, Cin, Cout, L, K, scale, shift, int(pre_relu), int(post_relu),
            None if res1 is None else res1.contiguous(), None if res2 is None else res2.contiguous(), T._stream(x))
    return y


def _relu(x):
    y = torch.empty_like(x)
    T._call("mural_op_act_fwd", x, x.numel(), 1, y, T._stream(x))
    return y


def _pool(x, k, s, p):
    x = x.contiguous()
    B, Cn, L = x.shape
    Lout = (L + 2 * p - k) // s + 1
    y = torch.empty((B, Cn, Lout), device=x.device)
    T._call("mural_op_maxpool_fwd", x, B * Cn, L, k, s, p, y, None, T._stream(x))      # (no argmax: nothing is differentiated here)
    return y


def _linear(x, lin):
    x = x.contiguous()
    y = torch.empty((x.shape[0], lin.out_features), device=x.device)
    T._call("mural_op_linear_fwd", x, T._f32(lin.weight.detach()), T._f32(lin.bias.detach()), x.shape[0], lin.in_features,
            lin.out_features, y, T._stream(x))
    return y


def _res_blocks(rbs, x):
    out = x
    n = len(rbs)
    for i, rb in enumerate(rbs):         # ResBlock (model_snv.py:794-812): x + conv2(bn2(relu(conv1(bn1(relu(x))))))
        h = _bn_conv(out, rb.bn1, True, rb.conv1)
        # the block's own skip rides in the second conv's epilogue, and so does the outer skip (model_snv.py:477-479) on the last block
        out = _bn_conv(h, rb.bn2, True, rb.conv2, res1=out, res2=x if i == n - 1 else None)
    return out if n else x + x


def tower(mod, sfx, x, pools):
    g = lambda n: getattr(mod, n + sfx)  # noqa: E731
    out = _pool(_bn_conv(x, g("conv1")[0], False, g("conv1")[1]), *pools[0])
    out = _pool(_res_blocks(g("RBs1"), out), *pools[1])
    out = _bn_conv(out, g("conv2")[0], False, g("conv2")[1])
    out = _pool(_res_blocks(g("RBs2"), out), *pools[2])
    assert out.shape[2] >= 1, "Error: distal seq is too short for the pooling layers"
    out = _bn_conv(out, g("conv3")[0], False, g("conv3")[1], post_relu=True)
    feat = _pool(out, out.shape[2], out.shape[2], 0).reshape(out.shape[0], out.shape[1])
    fc = mod.distal_fc1 if sfx == "" else mod.distal_fc2
    return _linear(_bn(feat, fc[0], False), fc[2])


def tower_tail(mod, sfx, s3, pools):
    """The tower from its second conv stage on: `s3` (B, C, L3) = conv2's BatchNorm of the pooled output of RBs1 (model_snv.py:477-483:
    the fused kernels apply that BatchNorm when they write the pooled row), as ``mural_snv_forward_front`` hands it out for long windows."""
    g = lambda n: getattr(mod, n + sfx)  # noqa: E731
    out = _bn_conv(s3, _NoBn(s3.shape[1], s3.device), False, g("conv2")[1])
    out = _pool(_res_blocks(g("RBs2"), out), *pools[2])
    assert out.shape[2] >= 1, "Error: distal seq is too short for the pooling layers"
    out = _bn_conv(out, g("conv3")[0], False, g("conv3")[1], post_relu=True)
    feat = _pool(out, out.shape[2], out.shape[2], 0).reshape(out.shape[0], out.shape[1])
    fc = mod.distal_fc1 if sfx == "" else mod.distal_fc2
    return _linear(_bn(feat, fc[0], False), fc[2])


def forward_from_front(mod, cat_x, mid_x, s3_large, pools_mid, pools_large):
    """log-probabilities from the fused front of the large tower (`s3_large` (B, C, L3)), the mid tower's 201-column window `mid_x`
    (B, 4, 201) and the k-mer ids `cat_x` (None: Network1)."""
    with torch.no_grad():
        mid = tower(mod, "", mid_x.to(torch.float32).contiguous(), pools_mid)
        lar = tower_tail(mod, "_2", s3_large, pools_large)
        loc = None if cat_x is None else local(mod, cat_x, mod.local_fc[0])
        return head(loc, mid, lar)


def local(mod, cat, out_layer):
    cat = cat.contiguous()
    table = T._f32(mod.emb_layer.weight.detach())
    h = torch.empty((cat.shape[0], cat.shape[1] * table.shape[1]), device=table.device)
    if table.shape[1] != 5:
        raise ValueError("the embedding kernel is built for 5-dimensional k-mer embeddings")
    T._call("mural_op_embedding_fwd", cat, table, cat.shape[0], cat.shape[1], table.shape[0], h, T._stream(table))
    for lin, bn in zip(mod.lin_layers, mod.bn_layers):
        h = _bn(_linear(h, lin), bn, True)          # Linear -> ReLU -> BN (model_snv.py:466-467)
    return _linear(h, out_layer)


def head(loc, mid, lar):
    B, nc = mid.shape
    out = torch.empty((B, nc), device=mid.device)
    T._call("mural_op_head_fwd", None if loc is None else loc.contiguous(), mid.contiguous(), lar.contiguous(), B, nc, out,
            T._stream(mid))
    return out


def forward(mod, cat_x, distal_x, pools_mid, pools_large):
    """log-probabilities of Network1 (cat_x None) / Network2 from the reference's dense inputs"""
    with torch.no_grad():
        L = distal_x.shape[2]
        x = distal_x.to(torch.float32)
        mid = tower(mod, "", x[:, :, L // 2 - 100:L // 2 + 101].contiguous(), pools_mid)
        lar = tower(mod, "_2", x.contiguous(), pools_large)
        loc = None if cat_x is None else local(mod, cat_x, mod.local_fc[0])
        return head(loc, mid, lar)
