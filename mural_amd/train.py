"""Whole-step hipGraph replay of the training step (``GraphedTrainStep``: SNV models, ``GraphedIndelTrainStep``: UNet_Small).

The eager step (MuRaL/training.py:424-436: forward, CE-sum loss, zero_grad, backward, clip_grad_norm_, optimizer.step) is
about 300 kernel launches behind ~4 ms of Python; once the kernels are fast the host is the bottleneck.  ``GraphedTrainStep``
captures one step into a HIP graph (``torch.cuda.CUDAGraph``) and replays it: the host then costs one launch per step.

Differences from a plain loop, all of them the usual graph-capture contract:
  * batch shapes are fixed at capture; inputs are copied into static tensors before every replay;
  * the optimizer must be capturable (``torch.optim.Adam(..., capturable=True)``) and is stepped inside the graph;
  * dropout masks: the host-drawn seeds are baked into the graph, a device-resident counter advanced inside the graph is
    added to them, so every replay draws new masks (``train_ops.set_device_seed``);
  * the encoding check of ``distal_x`` (ValueError on a non-MuRaL column) is read back after the replay, one step late.

The INDEL step (MuRaL/training.py:404-450 over model_indel.py:151-176) is ~500 small launches behind 7-11 ms of Python autograd
glue, depending on the host; its replay is bound by the device alone.
"""
import torch

from .model import train_ops as T

from ._host import freeze_host_heap  # noqa: E402,F401  (re-exported: train_epoch's loop and the benches call it)


class GraphedTrainStep:
    def __init__(self, model, optimizer, criterion, cont_x, cat_x, distal_x, y, max_norm=10.0, warmup=3):
        self.cont, self.cat = (None if t is None else t.clone() for t in (cont_x, cat_x))
        self._capture(model, optimizer, criterion, distal_x, y, max_norm, warmup)

    def _capture(self, model, optimizer, criterion, distal_x, y, max_norm, warmup):
        dev = distal_x.device
        self.model, self.opt, self.crit, self.max_norm = model, optimizer, criterion, max_norm
        self.x, self.y = distal_x.clone(), y.clone()
        self.seed = torch.zeros(1, dtype=torch.int64, device=dev)
        self.loss = None
        self._status = []
        self._pending = None
        T.set_device_seed(self.seed)
        try:
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                for _ in range(warmup):        # allocator warm-up, lazy kernel attributes, optimizer state
                    self._eager_step()
            torch.cuda.current_stream(dev).wait_stream(side)
            torch.cuda.synchronize(dev)
            T.reset_zero_arena()
            T.captured_status.clear()
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self._eager_step()
            self._status = list(T.captured_status)
            T.captured_status.clear()
            T.reset_zero_arena()
        finally:
            T.set_device_seed(None)

    def _eager_step(self):
        self.seed += 0x9E3779B97F4A7C15 & 0x7FFFFFFFFFFFFFFF       # new dropout masks every step (odd 63-bit increment)
        out = self._forward()
        self.loss = self.crit(out, self.y)
        # gradients of the one-call HIP step are views of a buffer that lives with the model (model/train_step.py: direct-gradient
        # mode): dropped here, set again by the backward -- same addresses every step, nothing to zero or to accumulate into
        self.opt.zero_grad(set_to_none=True)
        self.loss.backward()
        clip_grad_norm_(self.model, self.max_norm)
        self.opt.step()

    def _forward(self):
        return self.model((self.cont, self.cat), self.x)

    def __call__(self, cont_x, cat_x, distal_x, y):
        """Run one training step on this batch; returns the (device) loss tensor of the step."""
        self.cont.copy_(cont_x, non_blocking=True)
        self.cat.copy_(cat_x, non_blocking=True)
        return self._replay(distal_x, y)

    def _replay(self, distal_x, y):
        self._check()
        self.x.copy_(distal_x, non_blocking=True)
        self.y.copy_(y, non_blocking=True)
        self.graph.replay()
        if self._status:
            host = torch.empty(len(self._status), dtype=torch.int32, pin_memory=True)
            for i, s in enumerate(self._status):
                host[i:i + 1].copy_(s, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            self._pending = (ev, host)
        return self.loss

    def _check(self):
        if self._pending is not None:
            ev, host = self._pending
            ev.synchronize()
            self._pending = None
            if int(host.max()) != 0:
                raise ValueError("distal_input of the previous step held a column that is not a MuRaL one-hot / IUPAC-fraction "
                                 "encoding")

    def finish(self):
        """Wait for the last replay and raise its deferred input check, if any."""
        torch.cuda.synchronize()
        self._check()


class GraphedIndelTrainStep(GraphedTrainStep):
    """The same for ``UNet_Small`` (one input tensor): ``step = GraphedIndelTrainStep(model, opt, crit, x, y); loss = step(x, y)``."""

    def __init__(self, model, optimizer, criterion, x, y, max_norm=10.0, warmup=3):
        self._capture(model, optimizer, criterion, x, y, max_norm, warmup)

    def _forward(self):
        return self.model(self.x)

    def __call__(self, x, y):
        return self._replay(x, y)


class _CESum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y):
        from . import _lib
        B, nc = x.shape
        prob = torch.empty_like(x)
        loss = torch.empty((), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(_lib.lib().mural_op_ce_sum_fwd(x.data_ptr(), y.data_ptr(), B, nc, prob.data_ptr(), loss.data_ptr(),
                                                     _lib.current_stream_ptr(x.device)))
        ctx.save_for_backward(prob, y)
        return loss

    @staticmethod
    def backward(ctx, g):
        from . import _lib
        prob, y = ctx.saved_tensors
        B, nc = prob.shape
        g = g.to(torch.float32).contiguous()
        dx = torch.empty_like(prob)
        with torch.cuda.device(prob.device):
            _lib.check(_lib.lib().mural_op_ce_sum_bwd(prob.data_ptr(), y.data_ptr(), g.data_ptr(), B, nc, dx.data_ptr(),
                                                     _lib.current_stream_ptr(prob.device)))
        return dx, None


class CrossEntropySum(torch.nn.Module):
    """``torch.nn.CrossEntropyLoss(reduction='sum')`` (training.py:327, the criterion of the reference's loops) as one launch per
    direction: between the model's forward and backward torch's version is a chain of four to six dependent little launches (log-softmax,
    gather / reduce, their backwards) with nothing else to run beside them.  Same value up to the order of the sum (fixed here: the loss
    is reproducible), same gradient; a label outside [0, n_class) gives NaN where torch raises a device assert.  Anything but a float32
    (B, n_class) device tensor with int64 labels of at most 65536 rows goes to torch's implementation."""

    def forward(self, x, y):
        if (x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and y.dim() == 1 and y.dtype == torch.int64 and y.is_cuda
                and x.shape[0] == y.shape[0] and 0 < x.shape[0] <= 65536):
            return _CESum.apply(x.contiguous(), y.contiguous())
        return torch.nn.functional.cross_entropy(x, y, reduction="sum")


def clip_grad_norm_(model, max_norm):
    """``torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm)`` (training.py:430) for a model whose gradients came out of
    the one-call HIP backward: they are views of ONE flat buffer (zero between the slots), so the total 2-norm is one reduction
    over it and the scaling one multiply -- three launches instead of a 150-tensor foreach.  Any other state (a gradient that
    was replaced or accumulated elsewhere, a model without the flat buffer) falls through to torch's implementation."""
    lay = getattr(model, "_train_layout", None)
    flat = getattr(lay, "last_flat", None) if lay is not None else None
    if flat is not None:
        base = flat.data_ptr()
        for p, off in zip(lay.plist, lay.poffs):        # every gradient still sits in its slot of the latest backward's buffer
            g = p.grad
            if g is None or g.data_ptr() != base + 4 * off:
                break
        else:
            # two launches of the library (sums of squares per workgroup in double, then coefficient + scaling by every workgroup) instead of
            # torch's single-workgroup reduction and five scalar launches: 50 -> 12 us at the tail of every step
            from . import _lib
            scratch = getattr(lay, "clip_scratch", None)
            if scratch is None or scratch.device != flat.device:
                scratch = lay.clip_scratch = torch.empty(64, dtype=torch.float64, device=flat.device)
            total = torch.empty((), dtype=torch.float32, device=flat.device)      # (a fresh tensor per call: the caller may keep it)
            _lib.check(_lib.lib().mural_op_clip_grad_norm(flat.data_ptr(), flat.numel(), float(max_norm), scratch.data_ptr(),
                                                          total.data_ptr(), _lib.current_stream_ptr(flat.device)))
            return total
    # separate gradient tensors (UNet_Small, the per-layer SNV paths): torch's own sequence of launches -- per-tensor norms by one
    # foreach call, the norm of those, one foreach multiply -- without walking the module tree and regrouping ~150 tensors every
    # step (0.3 of the 0.8 ms that call costs on the host; the parameter list is cached on the model)
    plist = getattr(model, "_clip_plist", None)
    if plist is None:
        plist = model._clip_plist = [p for p in model.parameters()]
    grads = [p.grad for p in plist if p.grad is not None]
    if not grads:
        return torch.zeros((), device=plist[0].device if plist else "cpu")
    dev, dt = grads[0].device, grads[0].dtype
    if any(g.device != dev or g.dtype != dt for g in grads):
        return torch.nn.utils.clip_grad_norm_(plist, max_norm=max_norm, error_if_nonfinite=False)
    total = torch.linalg.vector_norm(torch.stack(torch._foreach_norm(grads, 2.0)), 2.0)
    torch._foreach_mul_(grads, torch.clamp(max_norm / (total + 1e-6), max=1.0))
    return total


class Adam(torch.optim.Adam):
    """``torch.optim.Adam`` (the reference's default optimiser, training.py:346-350, stepped at :432) with a ONE-launch step for a model
    whose training step runs in the library (model/train_step.py): there every gradient is a view of one flat buffer, so the
    parameters and the two moment buffers are laid out the same way -- ``p.data`` becomes a view of one flat buffer at the first step,
    ``state[p]['exp_avg']`` / ``['exp_avg_sq']`` are views as well -- and the update is one elementwise kernel
    (``mural_op_adam_flat``: torch's fused update rule, bias corrections from the host) instead of torch's multi-tensor sequence (a
    step-counter foreach + three launches over ~150 tensor quadruples, 40 us at the end of every step with nothing beside them).

    Same constructor as ``torch.optim.Adam``.  Everything the one launch does not cover goes through ``torch.optim.Adam.step`` itself,
    on the same state tensors: more than one parameter group, amsgrad / maximize / capturable / differentiable, a tensor learning rate,
    a closure, parameters that are not exactly one HIP model's (or a model without the flat gradient buffer: UNet_Small, CPU), a gradient
    that is not the latest backward's view.  ``state_dict`` / ``load_state_dict`` are torch's (step counters are materialised first)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False, **kw):
        params = list(params)
        if "fused" not in kw and "foreach" not in kw:      # torch's fused implementation for the steps that fall through, where it applies
            flat = [q for p in params for q in (p["params"] if isinstance(p, dict) else [p])]
            kw["fused"] = bool(flat) and all(torch.is_tensor(q) and q.is_cuda and q.is_floating_point() for q in flat)
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=amsgrad, **kw)
        self._flat = None          # (layout, params, param buffer, exp_avg buffer, exp_avg_sq buffer) while the one-launch step applies
        self._flat_t = 0           # updates done (the step counters of torch's state, materialised on demand)

    # -- torch's state <-> the flat buffers -----------------------------------------------------------------------------------
    def _materialise_steps(self):
        if self._flat is None:
            return
        g = self.param_groups[0]
        on_dev = bool(g.get("fused")) or bool(g.get("capturable"))
        for p in self._flat[1]:
            self.state[p]["step"] = (torch.full((), float(self._flat_t), dtype=torch.float32, device=p.device) if on_dev
                                     else torch.tensor(float(self._flat_t), dtype=torch.float32))

    def _leave_flat(self):
        self._materialise_steps()
        self._flat = None

    def _enter_flat(self, lay, params):
        dev = params[0].device
        if any(p.dtype is not torch.float32 or p.device != dev or not p.is_cuda for p in params):
            return False
        have = [p for p in params if len(self.state.get(p, {}))]
        t = 0
        if have:
            if len(have) != len(params):
                return False
            steps = torch.stack([self.state[p]["step"].detach().to(device=dev, dtype=torch.float32).reshape(()) for p in params])
            t = float(steps[0].item())
            if t != int(t) or not bool((steps == t).all().item()):
                return False
        pf, mf, vf = (torch.zeros(lay.total, dtype=torch.float32, device=dev) for _ in range(3))
        for p, o in zip(lay.plist, lay.poffs):
            n = p.numel()
            slot = lambda f: f[o:o + n].view(p.shape)      # noqa: E731
            slot(pf).copy_(p.data)
            st = self.state[p]
            if have:
                slot(mf).copy_(st["exp_avg"])
                slot(vf).copy_(st["exp_avg_sq"])
            p.data = slot(pf)
            st["exp_avg"], st["exp_avg_sq"] = slot(mf), slot(vf)
        self._flat, self._flat_t = (lay, list(lay.plist), pf, mf, vf), int(t)
        self._materialise_steps()
        return True

    def _flat_step(self):
        if len(self.param_groups) != 1:
            return False
        g = self.param_groups[0]
        if (g.get("amsgrad") or g.get("maximize") or g.get("capturable") or g.get("differentiable") or torch.is_tensor(g["lr"])
                or g.get("decoupled_weight_decay")):
            return False
        if self._flat is None:
            params = [p for p in g["params"] if p.numel()]
            from .model.train_step import layout_of_params
            lay = layout_of_params(params)
            if lay is None or lay.last_flat is None or not self._enter_flat(lay, params):
                return False
        lay, params, pf, mf, vf = self._flat
        flat = lay.last_flat
        if flat is None or flat.device != pf.device or flat.numel() != pf.numel():
            return False
        gbase, pbase = flat.data_ptr(), pf.data_ptr()
        for p, off in zip(params, lay.poffs):
            gr = p.grad
            if gr is None or gr.data_ptr() != gbase + 4 * off or p.data_ptr() != pbase + 4 * off:
                return False
        from . import _lib
        b1, b2 = g["betas"]
        self._flat_t += 1
        with torch.cuda.device(pf.device):
            _lib.check(_lib.lib().mural_op_adam_flat(pf.data_ptr(), gbase, mf.data_ptr(), vf.data_ptr(), pf.numel(), float(g["lr"]), float(b1),
                                                     float(b2), float(g["eps"]), float(g["weight_decay"]), self._flat_t,
                                                     _lib.current_stream_ptr(pf.device)))
        model = lay.model_ref()
        if model is not None and not model.training:      # (a step in eval mode: the kernel bumps no tensor version the folded copy watches)
            model.invalidate_folded()
        return True

    @torch.no_grad()
    def step(self, closure=None):
        # (torch wraps the step of every optimizer class -- this one too -- with its step hooks: they fire around either route; the
        # parent's step is called without ITS wrapper so that they do not fire twice)
        if closure is None and self._flat_step():
            return None
        self._leave_flat()
        parent = torch.optim.Adam.step
        if getattr(parent, "hooked", False):
            parent = getattr(parent, "__wrapped__", parent)
        return parent(self, closure)

    def state_dict(self):
        self._materialise_steps()
        return super().state_dict()

    def load_state_dict(self, state_dict):
        self._flat = None            # the loaded tensors replace the views; the next step lays them out again
        return super().load_state_dict(state_dict)


# ------------------------------------------------------------------------------------------------------------------
# the reference's epoch loop (MuRaL/training.py:346-450) around the HIP models: optimiser / scheduler choice and the
# per-batch policy (skip batches of one row, clip at 10, step the scheduler every batch, restart a learning rate that decayed
# below min_lr).  Host logic only; the model does the device work.
# ------------------------------------------------------------------------------------------------------------------
def make_optimizer(config, params):
    """training.py:346-360: Adam / AdamW / AdamW2 (both amsgrad) / SGD (momentum 0.98, nesterov) by ``config['optim']``."""
    params = [p for p in params if p.requires_grad]
    lr, wd = config["learning_rate"], config["weight_decay"]
    name = config["optim"]
    if name == "Adam":
        return Adam(params, lr=lr, weight_decay=wd)      # torch.optim.Adam; one launch per step behind the library's training step
    if name in ("AdamW", "AdamW2"):
        return torch.optim.AdamW(params, lr=lr, weight_decay=wd, amsgrad=True)
    if name == "SGD":
        return torch.optim.SGD(params, lr=lr, weight_decay=wd, momentum=0.98, nesterov=True)
    raise ValueError(f"Error: unsupported optimization method {name}")


def make_scheduler(config, optimizer, train_size=None):
    """training.py:364-373: 'StepLR' (step_size (5000 * 128) // batch_size, gamma LR_gamma), 'StepLR2' (per-step decay from
    restart_lr to min_lr over one epoch), 'ROP' (ReduceLROnPlateau)."""
    kind = config["lr_scheduler"]
    if kind == "StepLR":
        return torch.optim.lr_scheduler.StepLR(optimizer, step_size=(5000 * 128) // config["batch_size"], gamma=config["LR_gamma"])
    if kind == "StepLR2":
        gamma = (config["min_lr"] / config["restart_lr"]) ** (1 / (train_size // config["batch_size"]))
        return torch.optim.lr_scheduler.StepLR(optimizer, step_size=1, gamma=gamma)
    if kind == "ROP":
        return torch.optim.lr_scheduler.ReduceLROnPlateau(optimizer, mode="min", factor=0.2, patience=1, threshold=0.0001, min_lr=1e-7)
    raise ValueError(f"unknown lr_scheduler {kind}")


def train_epoch(model, batches, criterion, optimizer, scheduler, config, device, model_type="snv", epoch=0):
    """One epoch of training.py:392-450 over an iterable of ``(y, cont_x, cat_x, distal_x)`` batches (the order
    ``generate_data_batches`` yields them).  Returns the summed loss."""
    model.train()
    freeze_host_heap()
    if epoch > 0 and config["lr_scheduler"] == "StepLR2":
        for g in optimizer.param_groups:
            g["lr"] = config["restart_lr"]
    total_loss = 0.0
    for y, cont_x, cat_x, distal_x in batches:
        if y.shape[0] == 1:                       # a single row cannot feed a batch-statistics BatchNorm (training.py:415)
            continue
        cat_x, cont_x, distal_x, y = cat_x.to(device), cont_x.to(device), distal_x.to(device), y.to(device)
        preds = model.forward((cont_x, cat_x), distal_x) if model_type == "snv" else model.forward(distal_x)
        loss = criterion(preds, y.long().squeeze())
        optimizer.zero_grad()
        loss.backward()
        clip_grad_norm_(model, 10)
        optimizer.step()
        total_loss += loss.item()
        if config["lr_scheduler"] != "ROP":
            scheduler.step()
            if optimizer.param_groups[0]["lr"] < config["min_lr"]:      # avoid very small learning rates
                for g in optimizer.param_groups:
                    g["lr"] = config["restart_lr"]
    return total_loss
