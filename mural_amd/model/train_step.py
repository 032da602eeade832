"""The SNV training step as ONE autograd node over two C calls (include/mural_hip.h: mural_snv_train_forward / _backward).

``loss.backward()`` of the reference (MuRaL/training.py:424-427) walks ~120 autograd nodes; here the whole model forward is one
``torch.autograd.Function`` whose forward and backward are one library call each -- the composition of the ~100 kernels per
direction lives in C++ (csrc/snv_train.hip), so the host costs two ctypes transitions per step instead of one per layer.
Serves the shipped shape (32 channels, kernel 3); other shapes train on the per-layer ops of ``train_ops.py`` / ``indel_train.py``.

Gradients: the C backward writes every parameter gradient into ONE flat buffer.  By default the node then sets ``p.grad`` itself
-- views of a buffer that lives with the model, created once -- and hands autograd nothing: routing ~150 tensors through the engine
(one AccumulateGrad node and two view constructions per parameter) cost 1.0 ms of host time per step, as much as the whole device
backward at batch 4096.  ``p.grad`` after ``loss.backward()`` is what the reference loop (training.py:424-436: zero_grad, backward,
clip_grad_norm_, step) reads; an existing ``p.grad`` is accumulated into, like autograd does.  What this skips is autograd's own
bookkeeping per parameter: tensor hooks on parameters, ``torch.autograd.grad(loss, params)`` and DistributedDataParallel's reducer
do not see these gradients -- ``MURAL_TRAIN_AUTOGRAD_PARAMS=1`` (or ``model._autograd_params = True``) routes them through the
engine again.  The step falls back to that route BY ITSELF when a direct gradient could go unseen: a parameter carries a tensor hook
or a post-accumulate-grad hook, or a process group of more than one rank is initialised (DistributedDataParallel's reducer hangs off
the AccumulateGrad nodes this mode skips; ``north_star`` keeps training single-GPU, so nothing is lost).  ``torch.autograd.grad``
on parameters other than the anchor fails with autograd's own "not used in the graph" error.  The ``.grad`` views of one backward
are handed out again by the next one only while nobody else still holds them (reference count of the views): a caller who keeps
last step's gradients for accumulation or logging gets a fresh buffer instead of having them rewritten in place.
"""
import os
import sys
import weakref

import ctypes as C

import numpy as np
import torch

from .. import _lib

MOMENTUM = 0.1
_SLOTS = C.sizeof(_lib.MuralSnvParams) // 8          # the parameter struct is a flat run of pointers


_LAYOUT_OF = {}      # id(first parameter) -> (weak reference to that parameter, weak reference to the layout): mural_amd.train.Adam finds it


def layout_of_params(params):
    """The layout whose parameter list is exactly `params` (any order), or None."""
    if not params or not _LAYOUT_OF:
        return None
    for p in params:
        hit = _LAYOUT_OF.get(id(p))
        if hit is not None and hit[0]() is p:
            lay = hit[1]()
            if lay is not None and len(lay.plist) == len(params) and {id(q) for q in lay.plist} == {id(q) for q in params}:
                return lay
    return None


class _Layout:
    """Which tensor sits in which pointer slot of MuralSnvParams, found by building the struct once with slot tokens."""

    def __init__(self, model):
        from .model_snv import _HostParams
        tensors = []

        class Tok(_HostParams):
            def ptr(self, t):
                tensors.append(t)
                return C.c_void_p(len(tensors))            # token = 1-based index into `tensors`

        tok = Tok()
        params = _lib.MuralSnvParams()
        if model.model_no != 1:
            params.local = tok.local(model, model.output_layer if model.model_no == 0 else model.local_fc[0])
        if model.model_no != 0:
            params.mid, params.large = tok.tower(model, ""), tok.tower(model, "_2")
        raw = np.frombuffer(bytes(params), dtype=np.int64)
        assert raw.shape[0] == _SLOTS
        self.slot_tensor = [tensors[v - 1] if v else None for v in raw.tolist()]
        self.params_all = list(model.parameters())     # cached: walking the module tree costs ~0.2 ms per step
        self.plist = [p for p in self.params_all if p.numel()]
        self.model_ref = weakref.ref(model)
        if self.plist:
            for k in [k for k, (r, _) in _LAYOUT_OF.items() if r() is None]:      # (entries of models that are gone)
                del _LAYOUT_OF[k]
            _LAYOUT_OF[id(self.plist[0])] = (weakref.ref(self.plist[0]), weakref.ref(self))
        self.last_flat = None                          # the flat buffer behind the .grad views of the latest backward (clip_grad_norm_)
        self.own_flat = None                           # direct-gradient mode: the buffer and the views into it, created once
        self.own_views = None
        self.own_base = None                           # holder counts of own_flat / own_views when they were made (_holder_counts)
        index = {id(p): i for i, p in enumerate(self.plist)}
        # gradient slots: offset of each parameter in one flat float32 buffer (running statistics have none)
        offs, o = [], 0
        for p in self.plist:
            offs.append(o)
            o += (p.numel() + 63) // 64 * 64
        self.total = o
        self.poffs = offs
        self.grad_off = np.full(_SLOTS, -1, np.int64)
        for s, t in enumerate(self.slot_tensor):
            if t is not None and id(t) in index:
                self.grad_off[s] = offs[index[id(t)]] * 4
        self.has_grad = self.grad_off >= 0
        self.counters = [m.num_batches_tracked for m in model.modules()
                         if isinstance(m, torch.nn.BatchNorm1d) and m.num_batches_tracked is not None and m.weight.numel()]
        seen, uniq = set(), []
        for t in self.counters:                       # ResBlock registers its BatchNorms twice
            if id(t) not in seen:
                seen.add(id(t))
                uniq.append(t)
        self.counters = uniq

    def params_struct(self):
        for t in self.slot_tensor:
            if t is not None and (t.dtype is not torch.float32 or not t.is_contiguous()):
                raise RuntimeError("the HIP training step needs contiguous float32 parameters and buffers")
        raw = np.array([0 if t is None else t.data_ptr() for t in self.slot_tensor], dtype=np.int64)
        s = _lib.MuralSnvParams()
        C.memmove(C.byref(s), raw.ctypes.data, _SLOTS * 8)
        return s

    def grads_struct(self, base):
        raw = np.where(self.has_grad, self.grad_off + base, 0)
        s = _lib.MuralSnvParams()
        C.memmove(C.byref(s), raw.ctypes.data, _SLOTS * 8)
        return s


def _layout(model):
    lay = getattr(model, "_train_layout", None)
    if lay is None:
        lay = model._train_layout = _Layout(model)
    return lay


def supported(model):
    return model.model_no == 0 or (model.out_channels == 32 and model.kernel_size == 3)


def _make_shape(model):
    sh = getattr(model, "_train_shape", None)
    if sh is None:
        local = model.model_no != 1
        towers = model.model_no != 0
        sh = model._train_shape = _lib.MuralSnvShape(
            model.model_no, model.n_class, model.no_of_cat if local else 0, model.emb_layer.num_embeddings if local else 0,
            model.lin_layers[0].out_features if local else 0, model.lin_layers[1].out_features if local else 0,
            model.out_channels if towers else 32, model.kernel_size if towers else 3, model.seq_len if towers else 0, 1e-5)
    return sh


class ModelStep(torch.autograd.Function):
    """out = model(cat_x, distal_x) in training mode; backward returns the gradient of every parameter."""

    @staticmethod
    def forward(ctx, model, shape, cat_x, symbols, drops, seeds, seed_dev, direct, *params):
        lay = _layout(model)
        dev = params[0].device
        ctx.direct = direct
        B = (cat_x if cat_x is not None else symbols).shape[0]
        lib = _lib.lib()
        need = int(lib.mural_snv_train_workspace_bytes(C.byref(shape), B))
        if need == 0:
            _lib.check(_lib.MURAL_E_INVALID)
        ws = torch.empty(need, dtype=torch.uint8, device=dev)
        out = torch.empty((B, model.n_class), dtype=torch.float32, device=dev)
        ps = lay.params_struct()
        stream = _lib.current_stream_ptr(dev)
        _lib.check(lib.mural_snv_train_forward(C.byref(shape), C.byref(ps), None if cat_x is None else cat_x.data_ptr(), None,
                                               None if symbols is None else symbols.data_ptr(), B, drops.ctypes.data, seeds.ctypes.data,
                                               None if seed_dev is None else seed_dev.data_ptr(), MOMENTUM, out.data_ptr(), ws.data_ptr(),
                                               need, None, stream))
        if lay.counters:
            torch._foreach_add_(lay.counters, 1)
        ctx.model, ctx.shape, ctx.ws, ctx.args, ctx.params = model, shape, ws, (cat_x, drops, seeds, seed_dev, B), params
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dout):
        model, shape, ws = ctx.model, ctx.shape, ctx.ws
        if ws is None:
            raise RuntimeError("the SNV training step keeps its saved activations for ONE backward (retain_graph=True is not supported: "
                               "run the forward again)")
        cat_x, drops, seeds, seed_dev, B = ctx.args
        lay = _layout(model)
        dev = ws.device
        direct = ctx.direct and all(p.grad is None for p in lay.plist)     # (an existing .grad is accumulated into: a fresh buffer then)
        if direct:
            # Last step's gradients that somebody still holds (kept across zero_grad(set_to_none=True)) are not rewritten in place.
            # Three holders are looked for, each against the count recorded when the views were made: Python references to a view
            # object, C++ references to its TensorImpl, and -- what a derived tensor (`g.view(-1)`, `g[0]`, `g.detach()`) or any
            # other holder of the memory adds -- references to the buffer's storage.
            if lay.own_views is not None and not _views_are_ours(lay):
                lay.own_flat = lay.own_views = None
            if lay.own_flat is None or lay.own_flat.device != dev:
                lay.own_flat = torch.zeros(lay.total, dtype=torch.float32, device=dev)
                lay.own_views = [lay.own_flat[o:o + p.numel()].view(p.shape) for p, o in zip(lay.plist, lay.poffs)]
                lay.own_base = _holder_counts(lay)
            flat = lay.own_flat      # every slot is rewritten by the call below; the padding between the slots stays zero
        else:
            flat = torch.zeros(lay.total, dtype=torch.float32, device=dev)    # zero padding between the slots: the norm of the buffer
        lay.last_flat = flat                                                   # is the norm of the gradients
        ps, gs = lay.params_struct(), lay.grads_struct(flat.data_ptr())
        dout = dout.contiguous()
        _lib.check(_lib.lib().mural_snv_train_backward(C.byref(shape), C.byref(ps), C.byref(gs), None if cat_x is None else cat_x.data_ptr(),
                                                       dout.data_ptr(), B, drops.ctypes.data, seeds.ctypes.data,
                                                       None if seed_dev is None else seed_dev.data_ptr(), ws.data_ptr(), ws.numel(),
                                                       _lib.current_stream_ptr(dev)))
        ctx.ws = None
        if ctx.direct:
            if direct:
                for p, v in zip(lay.plist, lay.own_views):
                    p.grad = v
            else:
                for p, o in zip(lay.plist, lay.poffs):
                    g = flat[o:o + p.numel()].view(p.shape)
                    if p.grad is None:
                        p.grad = g
                    else:
                        p.grad.add_(g)
            return (None,) * (8 + len(ctx.params))
        grads = {id(p): flat[o:o + p.numel()].view(p.shape) for p, o in zip(lay.plist, lay.poffs)}
        return (None,) * 8 + tuple(grads[id(p)] if p.numel() else torch.zeros_like(p) for p in ctx.params)


def _holder_counts(lay):
    """(references to the storage of the gradient buffer, Python + C++ references of every view into it)."""
    storage = torch._C._storage_Use_Count(lay.own_flat.untyped_storage()._cdata)
    return storage, [(sys.getrefcount(v), v._use_count()) for v in lay.own_views]


def _views_are_ours(lay):
    """Nobody but the layout holds last step's gradient views or anything derived from them (counts as at their creation)."""
    return _holder_counts(lay) == lay.own_base


def _direct_is_safe(params):
    """Direct-gradient mode only where nothing but ``p.grad`` observes the gradients (module docstring)."""
    if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
        return False
    for p in params:
        if not p.requires_grad or p._backward_hooks or getattr(p, "_post_accumulate_grad_hooks", None):
            return False
    return True


def run(model, cat_x, distal_x):
    """Training-mode forward of `model` on (cat_x int64 (B, cols) | None, distal_x float (B, 4, L) | uint8 (B, L) symbols | None)."""
    from . import train_ops as T
    shape = _make_shape(model)
    symbols = None
    if distal_x is not None and distal_x.dtype == torch.uint8 and distal_x.dim() == 2:
        symbols = distal_x                              # the encoder's symbols (model_snv._symbol_windows): nothing to convert or check
    elif distal_x is not None:
        symbols = T.dense_to_symbols(distal_x)          # flags non-encodings; checked behind the launches (flush_input_checks)
    ps = [model.emb_dropout_layer.p, model.droput_layers[0].p, model.droput_layers[1].p] if model.model_no != 1 else [0.0, 0.0, 0.0]
    ps += [model.distal_fc1[1].p, model.distal_fc2[1].p] if model.model_no != 0 else [0.0, 0.0]
    drops = np.asarray(ps, dtype=np.float32)
    seeds = np.zeros(5, dtype=np.uint64)
    for i, p in enumerate(ps):                          # one draw per active dropout, in the order of the reference's forward
        if p > 0.0:
            seeds[i] = int(torch.randint(0, 2 ** 62, (1,)).item())
    params = _layout(model).params_all
    # direct-gradient mode (module docstring): one parameter anchors the node in the autograd graph, the node sets every .grad itself
    direct = (not getattr(model, "_autograd_params", False) and not os.environ.get("MURAL_TRAIN_AUTOGRAD_PARAMS")
              and _direct_is_safe(params))
    try:
        if direct:
            anchor = next(p for p in params if p.numel())
            out = ModelStep.apply(model, shape, cat_x, symbols, drops, seeds, T._device_seed, True, anchor)
        else:
            out = ModelStep.apply(model, shape, cat_x, symbols, drops, seeds, T._device_seed, False, *params)
        T.flush_input_checks()
    except BaseException:
        T._pending_checks.clear()                       # a failed forward must not leave its input check to the next one
        raise
    return out
