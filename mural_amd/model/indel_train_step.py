"""The INDEL training step as ONE autograd node over two C calls (include/mural_hip.h: mural_indel_train_forward / _backward).

``UNet_Small.forward`` under ``model.train()`` (MuRaL/model/model_indel.py:151-176) inside the step of MuRaL/training.py:424-436: the
per-unit composition of ``indel_train.py`` walks ~40 autograd nodes and issues ~340 launches from Python; here the composition lives
in C++ (csrc/indel_train_step.hip) and the host pays two ctypes transitions per step.  ``loss.backward()``, ``clip_grad_norm_`` and
``torch.optim`` work unchanged: the node returns one gradient per parameter, all views of one flat buffer.
"""
import ctypes as C

import numpy as np
import torch

from .. import _lib

MOMENTUM = 0.1
_SLOTS = C.sizeof(_lib.MuralIndelParams) // 8          # the parameter struct is a flat run of pointers


class _Layout:
    """Which tensor sits in which pointer slot of MuralIndelParams, found by building the struct once with slot tokens."""

    def __init__(self, model):
        tensors = []

        def tok(t):
            tensors.append(t)
            return C.c_void_p(len(tensors))               # token = 1-based index into `tensors`

        bn = lambda m: _lib.MuralBN(tok(m.weight), tok(m.bias), tok(m.running_mean), tok(m.running_var))      # noqa: E731
        aff = lambda m: _lib.MuralAffine(tok(m.weight), tok(m.bias))                                             # noqa: E731
        convbn = lambda conv, b: _lib.MuralConvBN(aff(conv), bn(b))                                              # noqa: E731
        block = lambda cb: _lib.MuralConvBlock(tok(cb.conv[0].weight), bn(cb.conv[1]), tok(cb.conv[3].weight), bn(cb.conv[4]))   # noqa: E731
        p = _lib.MuralIndelParams()
        if model.use_reverse:
            p.sym = convbn(model.conv[0], model.conv[1])
        for i in range(model.N_LEVELS):
            p.up_l[i] = convbn(model.uplblocks[i][0], model.uplblocks[i][1])
            p.up_b[i] = block(model.upblocks[i][0])
        for j in range(model.N_LEVELS - 1):
            p.down_l[j] = convbn(model.downlblocks[j][1], model.downlblocks[j][2])
            p.down_b[j] = block(model.downblocks[j][0])
        p.out1, p.out_bn, p.out2 = aff(model.out_conv[0]), bn(model.out_conv[1]), aff(model.out_conv[3])
        p.fc_bn, p.fc = bn(model.out_fc[0]), aff(model.out_fc[2])
        raw = np.frombuffer(bytes(p), dtype=np.int64)
        assert raw.shape[0] == _SLOTS
        self.slot_tensor = [tensors[v - 1] if v else None for v in raw.tolist()]
        self.params_all = list(model.parameters())
        self.plist = [q for q in self.params_all if q.numel()]
        self.last_flat = None
        index = {id(q): i for i, q in enumerate(self.plist)}
        offs, o = [], 0
        for q in self.plist:
            offs.append(o)
            o += (q.numel() + 63) // 64 * 64
        self.total, self.poffs = o, offs
        self.grad_off = np.full(_SLOTS, -1, np.int64)
        for s, t in enumerate(self.slot_tensor):
            if t is not None and id(t) in index:
                self.grad_off[s] = offs[index[id(t)]] * 4
        self.has_grad = self.grad_off >= 0
        # num_batches_tracked: +1 per BatchNorm application; the strand-symmetrising layer's BatchNorm is applied twice per forward
        self.counters = [m.num_batches_tracked for m in model.modules()
                         if isinstance(m, torch.nn.BatchNorm1d) and m.num_batches_tracked is not None]
        self.twice = model.conv[1].num_batches_tracked if model.use_reverse else None

    def _struct(self, raw):
        s = _lib.MuralIndelParams()
        C.memmove(C.byref(s), raw.ctypes.data, _SLOTS * 8)
        return s

    def params_struct(self):
        for t in self.slot_tensor:
            if t is not None and (t.dtype is not torch.float32 or not t.is_contiguous()):
                raise RuntimeError("the HIP training step needs contiguous float32 parameters and buffers")
        return self._struct(np.array([0 if t is None else t.data_ptr() for t in self.slot_tensor], dtype=np.int64))

    def grads_struct(self, base):
        return self._struct(np.where(self.has_grad, self.grad_off + base, 0))


def _layout(model):
    # the cached layout holds the parameter / buffer OBJECTS: replacing one of them (model.conv.weight = nn.Parameter(...)) must
    # rebuild it -- compared by identity, a few dozen `is` tests per call
    lay = getattr(model, "_train_layout", None)
    if lay is not None:
        now = list(model.parameters()) + list(model.buffers())
        if len(now) != len(lay.identity) or any(a is not b for a, b in zip(now, lay.identity)):
            lay = None
    if lay is None:
        lay = model._train_layout = _Layout(model)
        lay.identity = list(model.parameters()) + list(model.buffers())
    return lay


def _shape(model, length):
    return _lib.MuralIndelShape(model.n_class, model.out_channels, model.kernel_size, (C.c_int32 * 6)(*model.downsize),
                                int(bool(model.use_reverse)), int(length), 1e-5)


class ModelStep(torch.autograd.Function):
    """out = model(x) in training mode; backward returns the gradient of every parameter."""

    @staticmethod
    def forward(ctx, model, x, drop_p, seed, seed_dev, *params):
        lay = _layout(model)
        dev = x.device
        B, length = x.shape[0], x.shape[2]
        shape = _shape(model, length)
        lib = _lib.lib()
        need = int(lib.mural_indel_train_workspace_bytes(C.byref(shape), B))
        if need == 0:
            raise ValueError(f"input length {length} is not compatible with down_list {list(model.downsize)}")
        ws = torch.empty(need, dtype=torch.uint8, device=dev)
        out = torch.empty((B, model.n_class), dtype=torch.float32, device=dev)
        ps = lay.params_struct()
        _lib.check(lib.mural_indel_train_forward(C.byref(shape), C.byref(ps), x.data_ptr(), B, float(drop_p), int(seed),
                                                 None if seed_dev is None else seed_dev.data_ptr(), MOMENTUM, out.data_ptr(), ws.data_ptr(),
                                                 need, _lib.current_stream_ptr(dev)))
        if lay.counters:
            torch._foreach_add_(lay.counters, 1)
        if lay.twice is not None:
            lay.twice.add_(1)
        ctx.model, ctx.shape, ctx.ws, ctx.args, ctx.params = model, shape, ws, (x, drop_p, seed, seed_dev, B), params
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dout):
        model, shape, ws = ctx.model, ctx.shape, ctx.ws
        if ws is None:
            raise RuntimeError("the INDEL training step keeps its saved activations for ONE backward (retain_graph=True is not supported: "
                               "run the forward again)")
        x, drop_p, seed, seed_dev, B = ctx.args
        lay = _layout(model)
        dev = ws.device
        flat = torch.zeros(lay.total, dtype=torch.float32, device=dev)     # zero padding between the slots
        lay.last_flat = flat
        ps, gs = lay.params_struct(), lay.grads_struct(flat.data_ptr())
        dout = dout.contiguous()
        _lib.check(_lib.lib().mural_indel_train_backward(C.byref(shape), C.byref(ps), C.byref(gs), x.data_ptr(), dout.data_ptr(), B,
                                                         float(drop_p), int(seed), None if seed_dev is None else seed_dev.data_ptr(),
                                                         ws.data_ptr(), ws.numel(), _lib.current_stream_ptr(dev)))
        ctx.ws = None
        grads = {id(p): flat[o:o + p.numel()].view(p.shape) for p, o in zip(lay.plist, lay.poffs)}
        return (None,) * 5 + tuple(grads[id(p)] if p.numel() else torch.zeros_like(p) for p in ctx.params)


def run(model, x):
    """Training-mode forward of `model` (UNet_Small) on x float32 (B, 4, L)."""
    from . import train_ops as T
    p = float(model.out_fc[1].p)
    seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if p > 0.0 else 0     # torch's CPU generator: torch.manual_seed applies
    return ModelStep.apply(model, x, p, seed, T._device_seed, *_layout(model).params_all)
