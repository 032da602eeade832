"""Drop-in SNV model classes backed by the gfx950 HIP library.

Mirror of the reference's MuRaL/model/model_snv.py API surface: ``Network0`` (:97-108), ``Network1`` (:111-287),
``Network2`` (:290-525) with the same constructor signatures, sub-module names and therefore the same
``state_dict()`` keys/order (incl. the double registration inside ``ResBlock``, :794-812), so the shipped
checkpoints load with ``load_state_dict(strict=True)``.

The torch sub-modules here are *parameter containers only*: ``forward`` never runs them.  In eval mode it hands
the raw parameters to ``libmural_hip.so`` (which folds BN statistics, builds the first-layer 3-mer tables and the
MFMA weight fragments) and launches the fused kernels on the current HIP stream; in training mode it composes the
per-op HIP kernels of ``train_ops.py`` under autograd (batch-statistics BatchNorm, dropout, every backward).
There is no CPU path.
"""
import ctypes as C

import numpy as np
import torch
import torch.nn as nn

from .. import _lib

POOLS_MID = ((3, 3, 1), (3, 3, 1), (3, 3, 1))          # model_snv.py:356,361,371
POOLS_LARGE = ((15, 15, 7), (7, 7, 3), (3, 3, 1))      # model_snv.py:399,404,414


class ResBlock(nn.Module):
    """Parameter holder for the pre-activation residual unit (model_snv.py:794-812)."""

    def __init__(self, in_channels=32, kernel_size=3, stride=1, padding=0, dilation=1):
        super().__init__()
        mk = lambda: nn.Conv1d(in_channels, in_channels, kernel_size=kernel_size, stride=stride, padding=padding,
                               dilation=dilation)
        self.bn1 = nn.BatchNorm1d(in_channels)
        self.conv1 = mk()
        self.bn2 = nn.BatchNorm1d(in_channels)
        self.conv2 = mk()
        self.layer = nn.Sequential(nn.ReLU(), self.bn1, self.conv1, nn.ReLU(), self.bn2, self.conv2)


def _add_local(mod, emb_dims, no_of_cont, lin_layer_sizes, emb_dropout, lin_layer_dropouts, emb_padding_idx):
    if no_of_cont != 0:
        raise ValueError("continuous (bigWig) local features are not supported by the HIP path (n_cont must be 0)")
    mod.no_of_cat = len(emb_dims)
    mod.emb_layer = nn.Embedding(emb_padding_idx + 1, 5)
    mod.no_of_embs = len(emb_dims) * 5
    mod.no_of_cont = no_of_cont
    sizes = [mod.no_of_embs + no_of_cont] + list(lin_layer_sizes)
    mod.lin_layers = nn.ModuleList([nn.Linear(sizes[i], sizes[i + 1]) for i in range(len(sizes) - 1)])
    mod.first_bn_layer = nn.BatchNorm1d(no_of_cont)
    mod.bn_layers = nn.ModuleList([nn.BatchNorm1d(s) for s in lin_layer_sizes])
    mod.emb_dropout_layer = nn.Dropout(emb_dropout)
    mod.droput_layers = nn.ModuleList([nn.Dropout(p) for p in lin_layer_dropouts])
    if len(lin_layer_sizes) != 2:
        raise ValueError("the HIP local branch is built for two hidden layers (local_hidden1_size, local_hidden2_size)")


def _add_tower(mod, sfx, in_channels, out_channels, kernel_size, pools, dropout, n_class):
    pad = (kernel_size - 1) // 2
    bn_conv = lambda cin, relu=False: nn.Sequential(*([nn.BatchNorm1d(cin), nn.Conv1d(cin, out_channels, kernel_size, 1, pad)]
                                                      + ([nn.ReLU()] if relu else [])))
    rbs = lambda: nn.Sequential(*[ResBlock(out_channels, kernel_size=3, stride=1, padding=1, dilation=1) for _ in range(2)])
    setattr(mod, "conv1" + sfx, bn_conv(in_channels))
    setattr(mod, "maxpool1" + sfx, nn.MaxPool1d(*pools[0]))
    setattr(mod, "RBs1" + sfx, rbs())
    setattr(mod, "maxpool2" + sfx, nn.MaxPool1d(*pools[1]))
    setattr(mod, "conv2" + sfx, bn_conv(out_channels))
    setattr(mod, "RBs2" + sfx, rbs())
    setattr(mod, "maxpool3" + sfx, nn.MaxPool1d(*pools[2]))
    setattr(mod, "conv3" + sfx, bn_conv(out_channels, relu=True))
    setattr(mod, "distal_fc1" if sfx == "" else "distal_fc2",
            nn.Sequential(nn.BatchNorm1d(out_channels), nn.Dropout(dropout), nn.Linear(out_channels, n_class)))


# ---------------------------------------------------------------------------------------------------------------
# parameter hand-over to the C ABI
# ---------------------------------------------------------------------------------------------------------------
class _HostParams:
    """Keeps contiguous fp32 host copies alive while the C side reads them."""

    def __init__(self):
        self.keep = []

    def ptr(self, t):
        a = np.ascontiguousarray(t.detach().to("cpu", torch.float32).numpy())
        self.keep.append(a)
        return a.ctypes.data_as(C.c_void_p)

    def bn(self, m):
        return _lib.MuralBN(self.ptr(m.weight), self.ptr(m.bias), self.ptr(m.running_mean), self.ptr(m.running_var))

    def affine(self, m):
        return _lib.MuralAffine(self.ptr(m.weight), self.ptr(m.bias))

    def resblock(self, rb):
        return _lib.MuralResBlock(self.bn(rb.bn1), self.affine(rb.conv1), self.bn(rb.bn2), self.affine(rb.conv2))

    def tower(self, mod, sfx):
        g = lambda n: getattr(mod, n + sfx)
        fc = mod.distal_fc1 if sfx == "" else mod.distal_fc2
        t = _lib.MuralTower()
        t.bn_in, t.conv_in = self.bn(g("conv1")[0]), self.affine(g("conv1")[1])
        t.rbs1[0], t.rbs1[1] = self.resblock(g("RBs1")[0]), self.resblock(g("RBs1")[1])
        t.bn_mid, t.conv_mid = self.bn(g("conv2")[0]), self.affine(g("conv2")[1])
        t.rbs2[0], t.rbs2[1] = self.resblock(g("RBs2")[0]), self.resblock(g("RBs2")[1])
        t.bn_out, t.conv_out = self.bn(g("conv3")[0]), self.affine(g("conv3")[1])
        t.fc_bn, t.fc = self.bn(fc[0]), self.affine(fc[2])
        return t

    def local(self, mod, out_layer):
        l = _lib.MuralLocal()
        l.emb = self.ptr(mod.emb_layer.weight)
        l.lin[0], l.lin[1] = self.affine(mod.lin_layers[0]), self.affine(mod.lin_layers[1])
        l.bn[0], l.bn[1] = self.bn(mod.bn_layers[0]), self.bn(mod.bn_layers[1])
        l.out = self.affine(out_layer)
        return l


class _on_device:
    """``torch.cuda.device(dev)`` only when `dev` is not already current (the context manager costs ~10 us per call, which is
    a sixth of a 16-site forward)."""
    __slots__ = ("ctx",)

    def __init__(self, dev):
        self.ctx = None if torch.cuda.current_device() == (dev.index or 0) else torch.cuda.device(dev)

    def __enter__(self):
        if self.ctx is not None:
            self.ctx.__enter__()

    def __exit__(self, *exc):
        if self.ctx is not None:
            self.ctx.__exit__(*exc)
        return False


class _HipSnvBase(nn.Module):
    """Shared machinery: handle cache keyed on parameter versions, workspace, launches."""

    model_no = -1
    REUSE_MIN_SITES = 1024      # forward_packed_reuse with a density rule: smaller chunks take the per-window kernels

    def _hip_init(self):
        self._handle = None
        self._handle_key = None
        self._ws = None
        self._ws_rows = [0, 0]      # rows the workspace is known to hold for the packed / dense entry
        self._status = None
        self._status_host = None    # pinned copy of the encoding status of earlier calls (read one call late)
        self._status_event = None
        self._fused = None          # None: not probed yet; False: this shape takes the per-layer path (generic_eval.py)
        self._plist = None

    # -- description of this model for the C side ------------------------------------------------------------
    def _shape_and_params(self):
        raise NotImplementedError

    # The folded eval-mode copy (BatchNorm statistics folded into weights, first-layer tables, MFMA fragments) is rebuilt
    # when a parameter changed through torch (data_ptr / _version of the ~100 parameters) and on every event that can change
    # parameters or BatchNorm buffers behind torch's back: a train()/eval() transition (the training kernels and a replayed
    # HIP graph update weights and running statistics without bumping tensor versions), load_state_dict, .to()/_apply.
    def invalidate_folded(self):
        """Force the next eval-mode forward to rebuild the folded weights (call after changing parameters or buffers by hand,
        e.g. through ``.data``, which bumps no version counter)."""
        self._handle_key = None
        self._plist = None
        self._ws_rows = [0, 0]
        if getattr(self, "_fused", None) is False:      # the per-layer eval path keeps per-module caches of its own
            from . import generic_eval
            generic_eval.invalidate(self)

    def train(self, mode=True):
        if bool(mode) != self.training:     # a real transition; model.eval() on a model in eval mode keeps the folded copy
            self.invalidate_folded()
        return super().train(mode)

    def _storage_signature(self):
        return tuple((t.data_ptr(), t.dtype, t.device) for t in list(self.parameters()) + list(self.buffers()))

    def _apply(self, fn, *args, **kwargs):
        before = self._storage_signature()
        out = super()._apply(fn, *args, **kwargs)
        if self._storage_signature() != before:    # .to() / .cuda() that moved or cast something (a no-op .to(device) does not)
            self.invalidate_folded()
            self._train_layout = None              # tensor objects / storages were replaced
        return out

    def load_state_dict(self, *args, **kwargs):
        self.invalidate_folded()
        self._train_layout = None
        return super().load_state_dict(*args, **kwargs)

    def _state_key(self):
        # in-place updates through torch bump _version (parameters and BatchNorm buffers alike); `p.data = other` keeps the version
        # but changes data_ptr; storage moves through .to() go through _apply / load_state_dict (hooked above) as well
        if self._plist is None:
            # num_batches_tracked (the only integer buffers) does not enter the eval-mode arithmetic
            self._plist = list(self.parameters()) + [b for b in self.buffers() if b.is_floating_point()]
        return [(t.data_ptr(), t._version) for t in self._plist]

    def _get_handle(self):
        """Folded device copy of the parameters; must be called under ``torch.cuda.device(model device)``."""
        key = self._handle_key
        if self._handle is not None and key is not None:
            if key == self._state_key():
                return self._handle
        key = self._state_key()
        self._release()
        shape, params, keep = self._shape_and_params()
        h = C.c_void_p()
        _lib.check(_lib.lib().mural_snv_model_create(C.byref(shape), C.byref(params), C.byref(h)))
        del keep
        self._handle, self._handle_key = h, key
        return self._handle

    def _fused_ok(self):
        """False when the fused kernels are not built for this shape (CNN_out_channels != 32, CNN_kernel_size != 3, a window
        too long for LDS): the C side refuses the model and eval takes one HIP launch per layer instead.  A FRONT-ONLY handle
        (distal_radius from ~15000: the second conv stage fits no fused kernel) counts as not fused; ``_front_ok`` tells."""
        if self._fused is None:
            try:
                with torch.cuda.device(self._device()):
                    self._get_handle()
                self._fused = not self._handle_front_only()
            except ValueError:
                self._fused = False
        return self._fused

    def _handle_front_only(self):
        lay = (C.c_int32 * 16)()
        _lib.check(_lib.lib().mural_snv_tap_layout(self._handle, lay))
        self._front_L3, self._front_chunk, self._front_mid = int(lay[3]), int(lay[11]), bool(lay[12])
        return bool(lay[10])

    def _front_ok(self):
        """The handle exists but serves ``mural_snv_forward_front`` only: the packed entry then runs the large tower's segmented
        first stage fused and finishes per layer (generic_eval.forward_from_front)."""
        if self._fused_ok() or self.model_no == 0:
            return False
        try:
            with torch.cuda.device(self._device()):
                self._get_handle()
                return self._handle_front_only()
        except ValueError:
            return False

    def _release(self):
        if getattr(self, "_handle", None) is not None:
            _lib.lib().mural_snv_model_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    def _workspace(self, n, device, dense=True):
        ws = self._ws
        if ws is not None and ws.device == device and n <= self._ws_rows[int(dense)]:
            return ws                                  # big enough for this many rows already (sizes grow with n)
        need = int(_lib.lib().mural_snv_workspace_bytes(self._get_handle(), n, int(dense)))
        if ws is None or ws.numel() < need or ws.device != device:
            self._ws = None
            try:
                self._ws = ws = torch.empty(need, dtype=torch.uint8, device=device)
            except torch.OutOfMemoryError:
                # the full size keeps four chunks of pooled rows (one short-stage launch per tower, + 1 %); the library accepts the
                # one-chunk layout as well (include/mural_hip.h: mural_snv_workspace_bytes_min) -- same results
                need = int(_lib.lib().mural_snv_workspace_bytes_min(self._get_handle(), n, int(dense)))
                self._ws = ws = torch.empty(need, dtype=torch.uint8, device=device)
            self._ws_rows = [0, 0]
        self._ws_rows[int(dense)] = max(self._ws_rows[int(dense)], n)
        return ws

    # -- encoding status of the dense entry, read without draining the device --------------------------------
    _ENC_MSG = ("distal_input holds a column that is not a MuRaL one-hot / IUPAC-fraction encoding "
                "(preprocessing.py:758-772); the HIP path consumes sequence encodings only")

    def check_encoding(self, wait=True):
        """Raise ValueError if an earlier dense forward saw a column that is not a sequence encoding.  The kernels flag it
        in a device word and overwrite that call's output with NaN; the word travels to pinned host memory behind the launches
        and is looked at here: without blocking at the start of the next forward, blocking when `wait` (model_predict_m does
        that once at the end of its loop)."""
        if self._status is None:
            return
        if wait:                               # authoritative: the device word itself (drains the stream)
            bad = int(self._status.item()) != 0
            self._status_event = None
        else:
            ev = self._status_event
            if ev is None or not ev.query():
                return
            self._status_event = None
            bad = int(self._status_host[0]) != 0
        if bad:
            self._status_host.zero_()
            self._status.zero_()
            raise ValueError(self._ENC_MSG)

    def _device(self):
        return next(self.parameters()).device

    def _check_eval(self):
        if self.training:
            raise RuntimeError("internal: the fused inference kernels serve eval mode only")

    def _train_inputs(self, cat_x, distal_x):
        dev = self._device()
        if dev.type != "cuda":
            raise RuntimeError("mural_amd models run on a HIP device only: call model.to('cuda') first")
        if cat_x is not None:
            cat_x = _lib.require_cuda(cat_x, "cat_x").to(torch.int64).contiguous()
        if distal_x is not None:
            distal_x = _lib.require_cuda(distal_x, "distal_x")
        return cat_x, distal_x

    # -- dense entry (drop-in forward) ------------------------------------------------------------------------
    def _forward_dense(self, cat_x, distal_x, taps=None):
        """Eval-mode ``model(local_input, distal_input)``.

        ENCODING ERRORS ARE REPORTED LATE.  A `distal_input` column that is not a MuRaL one-hot / IUPAC-fraction encoding cannot be
        evaluated by the sequence kernels.  The call that holds it does not raise: its output rows are NaN, a sticky device word is
        set, and every later dense call on this model also returns NaN until the host has seen the word -- without blocking at the
        start of a later forward, or at once through ``model.check_encoding(wait=True)`` -- and raised ``ValueError``.  Loops that
        consume outputs without a later forward must end with ``check_encoding(wait=True)`` (``model_predict_m`` does); NaN
        in an output always means "check_encoding will raise", never a silent wrong value."""
        self._check_eval()
        dev = self._device()
        if dev.type != "cuda":
            raise RuntimeError("mural_amd models run on a HIP device only: call model.to('cuda') first")
        n = None
        cat_ptr = dist_ptr = None
        if cat_x is not None:
            cat_x = _lib.require_cuda(cat_x, "cat_x").to(torch.int64).contiguous()
            n = cat_x.shape[0]
            cat_ptr = cat_x.data_ptr()
        if distal_x is not None:
            distal_x = _lib.require_cuda(distal_x, "distal_x").to(torch.float32).contiguous()
            n = distal_x.shape[0]
            dist_ptr = distal_x.data_ptr()
        if self.model_no != 0 and taps is None and not self._fused_ok():
            from . import generic_eval
            with torch.cuda.device(dev):
                return generic_eval.forward(self, cat_x, distal_x, POOLS_MID, POOLS_LARGE)
        with _on_device(dev):
            handle = self._get_handle()
            out = torch.empty((n, self.n_class), dtype=torch.float32, device=dev)
            ws = self._workspace(max(n, 1), dev)
            stream = _lib.current_stream_ptr(dev)
            if taps is None:
                self.check_encoding(wait=False)
                if self._status is None or self._status.device != dev:
                    self._status = torch.zeros(1, dtype=torch.int32, device=dev)
                    self._status_host = torch.zeros(1, dtype=torch.int32).pin_memory()
                _lib.check(_lib.lib().mural_snv_forward_dense(handle, cat_ptr, dist_ptr, n, out.data_ptr(), ws.data_ptr(),
                                                             ws.numel(), self._status.data_ptr(), stream))
                if self.model_no != 0 and self._status_event is None and not torch.cuda.is_current_stream_capturing():
                    self._status_host.copy_(self._status, non_blocking=True)
                    self._status_event = torch.cuda.Event()
                    self._status_event.record()
            else:
                _lib.check(_lib.lib().mural_snv_debug_taps(handle, cat_ptr, dist_ptr, n, out.data_ptr(), ws.data_ptr(),
                                                          ws.numel(), taps.data_ptr(), taps.numel(), stream))
        return out

    # -- fused encode + forward from the packed genome --------------------------------------------------------
    def forward_packed(self, genome, pos, strand, local_radius=None, local_order=3):
        """log-probabilities for sites `pos` (int64, 0-based) / `strand` (uint8, 1 = '-') of a PackedGenome."""
        self._check_eval()
        dev = self._device()
        pos = _lib.require_cuda(pos, "pos").to(torch.int64).contiguous()
        strand = _lib.require_cuda(strand, "strand").to(torch.uint8).contiguous()
        n = pos.shape[0]
        if local_radius is None:
            local_radius = (getattr(self, "no_of_cat", 1) + local_order - 2) // 2
        if self.model_no != 0 and self._front_ok():
            # a window too long for the fused short stages of the large tower: stage 1 and that tower's first conv stage (86 % of the
            # arithmetic) run fused on segments of the pooled row, its remaining ten layers per layer (generic_eval.tower_tail), and --
            # where their launches fit -- the mid tower, the local branch and the head fused again around its logits
            from . import generic_eval
            lib = _lib.lib()
            with torch.cuda.device(dev):
                handle = self._get_handle()
                L3, step = self._front_L3, (self._front_chunk if self._front_mid else max(n, 1))
                g = genome.as_struct(dev)
                stream = _lib.current_stream_ptr(dev)
                ws = self._workspace(max(min(n, step), 1), dev, dense=False)      # (kept with the model: the finish call reads what the front call left)
                out = torch.empty((n, self.n_class), dtype=torch.float32, device=dev)
                for c0 in range(0, n, step):
                    p, st = pos[c0:c0 + step], strand[c0:c0 + step]
                    m = p.shape[0]
                    s3 = torch.empty((m, L3, 32), dtype=torch.float32, device=dev)
                    _lib.check(lib.mural_snv_forward_front(handle, C.byref(g), p.data_ptr(), st.data_ptr(), m, int(local_radius), int(local_order),
                                                          s3.data_ptr(), ws.data_ptr(), ws.numel(), stream))
                    s3 = s3.permute(0, 2, 1).contiguous()
                    if self._front_mid:
                        large = generic_eval.tower_tail(self, "_2", s3, POOLS_LARGE).contiguous()
                        _lib.check(lib.mural_snv_forward_finish(handle, large.data_ptr(), m, out[c0:c0 + m].data_ptr(), ws.data_ptr(), ws.numel(),
                                                               stream))
                    else:
                        cat = genome.encode_kmer(p, st, int(local_radius), int(local_order)) if self.model_no == 2 else None
                        out[c0:c0 + m] = generic_eval.forward_from_front(self, cat, genome.encode_onehot(p, st, 100), s3, POOLS_MID, POOLS_LARGE)
                return out
        if self.model_no != 0 and not self._fused_ok():
            from . import generic_eval
            with torch.cuda.device(dev):
                x = genome.encode_onehot(pos, strand, (self.seq_len - 1) // 2)
                cat = genome.encode_kmer(pos, strand, int(local_radius), int(local_order)) if self.model_no == 2 else None
                return generic_eval.forward(self, cat, x, POOLS_MID, POOLS_LARGE)
        with torch.cuda.device(dev):
            handle = self._get_handle()
            out = torch.empty((n, self.n_class), dtype=torch.float32, device=dev)
            ws = self._workspace(max(n, 1), dev, dense=False)
            g = genome.as_struct(dev)
            _lib.check(_lib.lib().mural_snv_forward_packed(handle, C.byref(g), pos.data_ptr(), strand.data_ptr(), n,
                                                          int(local_radius), int(local_order), out.data_ptr(),
                                                          ws.data_ptr(), ws.numel(), _lib.current_stream_ptr(dev)))
        return out

    def forward_symbols(self, cat_x, symbols):
        """Eval-mode forward for windows given as one MURAL_SYM_* byte per column (device uint8 (n, 2 * distal_radius + 1)): what
        ``model_predict_m`` feeds after classifying a host loader's one-hot windows on the host (``mural_host_dense_to_symbols``)."""
        self._check_eval()
        dev = self._device()
        if dev.type != "cuda":
            raise RuntimeError("mural_amd models run on a HIP device only: call model.to('cuda') first")
        if not self.symbols_entry_ok():
            raise RuntimeError("this model configuration evaluates through the per-layer path, which takes the dense distal_input")
        symbols = _lib.require_cuda(symbols, "symbols")
        if symbols.dtype != torch.uint8 or symbols.dim() != 2 or symbols.shape[1] != self.seq_len:
            raise ValueError(f"symbols must be a uint8 (n, {self.seq_len}) tensor")
        symbols = symbols.contiguous()
        n = symbols.shape[0]
        cat_ptr = None
        if self.model_no != 1:
            cat_x = _lib.require_cuda(cat_x, "cat_x").to(torch.int64).contiguous()
            cat_ptr = cat_x.data_ptr()
        with _on_device(dev):
            out = torch.empty((n, self.n_class), dtype=torch.float32, device=dev)
            ws = self._workspace(max(n, 1), dev, dense=False)
            _lib.check(_lib.lib().mural_snv_forward_symbols(self._get_handle(), cat_ptr, symbols.data_ptr(), n, out.data_ptr(), ws.data_ptr(),
                                                           ws.numel(), _lib.current_stream_ptr(dev)))
        return out

    def symbols_entry_ok(self):
        return self.model_no != 0 and self._fused_ok()

    def forward_packed_reuse(self, genome, pos, strand, local_radius=None, local_order=3, min_density=0.0, batch_sites=1 << 20,
                             return_reuse_count=False):
        """``forward_packed`` with cross-position reuse (csrc/snv_reuse.hip): for site lists that are dense along a chromosome the
        first conv stage of both towers is evaluated once per base and strand instead of once per window; per site only the
        pooled columns next to its window edges are recomputed.  Same rows, same order, probabilities equal to the per-window
        path within rounding (GPU tests: 1e-5).  Falls back to ``forward_packed`` (in batches of `batch_sites`) for models the
        reuse kernels do not cover and for chunks of the position axis with fewer than `min_density` sites per base, where the
        per-base rows cost more than they save (with a positive `min_density` a chunk also needs REUSE_MIN_SITES sites: below that the
        fixed launches of the row kernels dominate).  With `return_reuse_count` the result is ``(out, sites that took the reuse
        kernels)``."""
        self._check_eval()
        dev = self._device()
        pos = _lib.require_cuda(pos, "pos").to(torch.int64).contiguous()
        strand = _lib.require_cuda(strand, "strand").to(torch.uint8).contiguous()
        n = pos.shape[0]
        lib = _lib.lib()
        if local_radius is None:
            local_radius = (getattr(self, "no_of_cat", 1) + local_order - 2) // 2
        done = lambda out, used: (out, used) if return_reuse_count else out      # noqa: E731

        def per_window(p, st, dst=None):
            parts = [self.forward_packed(genome, p[r0:r0 + batch_sites], st[r0:r0 + batch_sites], local_radius, local_order)
                     for r0 in range(0, max(p.shape[0], 1), batch_sites)]
            res = parts[0] if len(parts) == 1 else torch.cat(parts)
            if dst is None:
                return res
            dst.copy_(res)
            return dst

        if self.model_no == 0 or n == 0 or not self._fused_ok():
            return done(per_window(pos, strand), 0)
        with torch.cuda.device(dev):
            handle = self._get_handle()
            if not lib.mural_snv_reuse_supported(handle):
                return done(per_window(pos, strand), 0)
            out = torch.empty((n, self.n_class), dtype=torch.float32, device=dev)
            span = int(lib.mural_snv_reuse_chunk_span())
            g = genome.as_struct(dev)
            stream = _lib.current_stream_ptr(dev)
            # one host round trip: bounds of the site list and which strands occur
            p_min, p_max, s_min, s_max = torch.stack([pos.min(), pos.max(), strand.min().to(torch.int64),
                                                      strand.max().to(torch.int64)]).tolist()

            def run(p, st, lo, hi, strands, dst):
                need = int(lib.mural_snv_reuse_workspace_bytes(handle, p.shape[0], hi - lo + 1, strands))
                if self._ws is None or self._ws.numel() < need or self._ws.device != dev:
                    self._ws = None
                    self._ws = torch.empty(need, dtype=torch.uint8, device=dev)
                    self._ws_rows = [0, 0]
                _lib.check(lib.mural_snv_forward_packed_reuse(handle, C.byref(g), p.data_ptr(), st.data_ptr(), p.shape[0], strands, lo, hi,
                                                              int(local_radius), int(local_order), dst.data_ptr(),
                                                              self._ws.data_ptr(), self._ws.numel(), stream))

            strands = (1 if s_min == 0 else 0) | (2 if s_max != 0 else 0)
            used = 0
            if p_max - p_min + 1 <= span:              # the common case: one chunk, rows of both strands resident, sites in any order
                if min_density > 0 and (n < min_density * (p_max - p_min + 1) or n < self.REUSE_MIN_SITES):
                    return done(per_window(pos, strand, out), 0)
                run(pos, strand, p_min, p_max, strands, out)
                used = n
            else:                                      # long spans: sort once, one call per chunk of the position axis
                p_sorted, perm = torch.sort(pos)
                s_sorted = strand[perm].contiguous()
                # chunk k: [edges[k], edges[k+1]); the last edge lies strictly behind p_max (also when p_max - p_min is a multiple
                # of span), so that the half-open chunks cover every site
                edges = torch.arange(p_min, p_max + span + 1, span, device=dev, dtype=torch.int64)
                cuts = torch.searchsorted(p_sorted, edges).tolist()
                res_sorted = torch.empty_like(out)
                for k in range(len(cuts) - 1):
                    lo, hi = cuts[k], cuts[k + 1]
                    if hi == lo:
                        continue
                    c_lo = p_min + k * span
                    c_hi = min(c_lo + span - 1, p_max)
                    if min_density > 0 and (hi - lo < min_density * (c_hi - c_lo + 1) or hi - lo < self.REUSE_MIN_SITES):
                        per_window(p_sorted[lo:hi], s_sorted[lo:hi], res_sorted[lo:hi])
                        continue
                    run(p_sorted[lo:hi], s_sorted[lo:hi], c_lo, c_hi, strands, res_sorted[lo:hi])
                    used += hi - lo
                out.index_copy_(0, perm, res_sorted)
        return done(out, used)

    def tap_layout(self):
        arr = (C.c_int32 * 16)()
        _lib.check(_lib.lib().mural_snv_tap_layout(self._get_handle(), arr))
        return list(arr)


def _shape(model_no, n_class, local_cols=0, emb_rows=0, h1=0, h2=0, channels=32, ksize=3, distal_len=0):
    return _lib.MuralSnvShape(model_no, n_class, local_cols, emb_rows, h1, h2, channels, ksize, distal_len, 1e-5)


def _train_towers(mod, distal_x):
    """(mid, large) tower logits in training mode: the MFMA / table kernels for the shipped 32-channel k=3 shape, the general
    per-layer ops (indel_train.py) for any other CNN_out_channels / CNN_kernel_size."""
    from . import train_ops as T
    if mod.out_channels == 32 and mod.kernel_size == 3:
        sym = T.dense_to_symbols(distal_x)
        mid = T.tower_forward(mod, "", sym, mod.seq_len // 2 - 100, 201, POOLS_MID, mod.distal_fc1[1].p)
        large = T.tower_forward(mod, "_2", sym, 0, mod.seq_len, POOLS_LARGE, mod.distal_fc2[1].p)
        return mid, large
    from .indel_train import snv_tower_forward_train
    x = distal_x.to(torch.float32)
    L = x.shape[2]
    mid = snv_tower_forward_train(mod, "", x[:, :, L // 2 - 100:L // 2 + 101], POOLS_MID, mod.distal_fc1[1].p)
    large = snv_tower_forward_train(mod, "_2", x, POOLS_LARGE, mod.distal_fc2[1].p)
    return mid, large


class FeedForwardNN(_HipSnvBase):
    """Local-only network body (model_snv.py:19-95)."""
    model_no = 0

    def __init__(self, emb_dims, no_of_cont, lin_layer_sizes, emb_dropout, lin_layer_dropouts, n_class, emb_padding_idx=None):
        super().__init__()
        self.n_class = n_class
        _add_local(self, emb_dims, no_of_cont, lin_layer_sizes, emb_dropout, lin_layer_dropouts, emb_padding_idx)
        self.output_layer = nn.Linear(lin_layer_sizes[-1], n_class)
        self._hip_init()

    def _shape_and_params(self):
        hp = _HostParams()
        params = _lib.MuralSnvParams()
        params.local = hp.local(self, self.output_layer)
        shape = _shape(0, self.n_class, self.no_of_cat, self.emb_layer.num_embeddings, self.lin_layers[0].out_features,
                       self.lin_layers[1].out_features)
        return shape, params, hp

    def forward(self, cont_data, cat_data):
        if self.training:     # one autograd node over the C training step (train_step.py / csrc/snv_train.hip)
            from . import train_step
            cat_data, _ = self._train_inputs(cat_data, None)
            with torch.cuda.device(self._device()):
                return train_step.run(self, cat_data, None)
        return self._forward_dense(cat_data, None)


class Network0(nn.Module):
    """Wrapper with the common ((cont, cat), distal) call signature (model_snv.py:97-108)."""

    def __init__(self, emb_dims, no_of_cont, lin_layer_sizes, emb_dropout, lin_layer_dropouts, n_class, emb_padding_idx=None):
        super().__init__()
        self.model = FeedForwardNN(emb_dims, no_of_cont, lin_layer_sizes, emb_dropout, lin_layer_dropouts, n_class,
                                   emb_padding_idx)
        self.n_class = n_class

    def forward(self, local_input, distal_input=None):
        cont_data, cat_data = local_input
        return self.model.forward(cont_data, cat_data)

    def forward_packed(self, genome, pos, strand, local_radius=None, local_order=3):
        return self.model.forward_packed(genome, pos, strand, local_radius, local_order)

    def forward_packed_reuse(self, genome, pos, strand, local_radius=None, local_order=3, min_density=0.0, batch_sites=1 << 20,
                             return_reuse_count=False):
        # no conv towers: nothing to share
        return self.model.forward_packed_reuse(genome, pos, strand, local_radius, local_order, min_density, batch_sites,
                                               return_reuse_count)


def _symbol_windows(model, distal_input):
    """``distal_input`` handed over as ``mural_amd.data.SymbolWindows`` (PackedGenome.encode_symbols): the uint8 (B, W) tensor for the
    training step, after the checks the dense route makes on its tensor; None for a dense tensor."""
    from ..data.genome import SymbolWindows
    if not isinstance(distal_input, SymbolWindows):
        return None
    sym = distal_input.sym
    if sym.dim() != 2 or sym.dtype != torch.uint8:
        raise TypeError("SymbolWindows.sym must be a uint8 (B, W) tensor")
    assert sym.shape[1] > 200, "Error: distal seq len must be >200bp"
    if sym.shape[1] != model.seq_len:
        raise ValueError(f"distal_input length {sym.shape[1]} != 2*distal_radius+1 = {model.seq_len}")
    if not model.training:
        raise TypeError("SymbolWindows are the training step's input; for prediction from a packed genome use forward_packed")
    if getattr(model, "in_channels", 4) != 4:
        raise ValueError("SymbolWindows stand for the four one-hot channels")
    from . import train_step
    if not train_step.supported(model):
        raise ValueError("this model configuration trains through the per-layer path, which takes the dense distal_input")
    return _lib.require_cuda(sym, "distal_input").contiguous()


class Network1(_HipSnvBase):
    """Expanded-only model (model_snv.py:111-287)."""
    model_no = 1

    def __init__(self, in_channels, out_channels, kernel_size, distal_radius, distal_order, distal_fc_dropout, n_class):
        super().__init__()
        self.n_class, self.in_channels, self.kernel_size = n_class, in_channels, kernel_size
        self.out_channels = out_channels
        self.seq_len = distal_radius * 2 + 1 - (distal_order - 1)
        _add_tower(self, "", in_channels, out_channels, kernel_size, POOLS_MID, distal_fc_dropout, n_class)
        _add_tower(self, "_2", in_channels, out_channels, kernel_size, POOLS_LARGE, distal_fc_dropout, n_class)
        self._hip_init()

    def _shape_and_params(self):
        hp = _HostParams()
        params = _lib.MuralSnvParams()
        params.mid, params.large = hp.tower(self, ""), hp.tower(self, "_2")
        shape = _shape(1, self.n_class, channels=self.out_channels, ksize=self.kernel_size, distal_len=self.seq_len)
        return shape, params, hp

    def forward(self, local_input, distal_input):
        sym = _symbol_windows(self, distal_input)
        if sym is not None:
            from . import train_step
            with torch.cuda.device(self._device()):
                return train_step.run(self, None, sym)
        assert distal_input.shape[2] > 200, "Error: distal seq len must be >200bp"
        if distal_input.shape[2] != self.seq_len:
            raise ValueError(f"distal_input length {distal_input.shape[2]} != 2*distal_radius+1 = {self.seq_len}")
        if self.training:
            from . import train_ops as T
            from . import train_step
            _, distal_input = self._train_inputs(None, distal_input)
            with torch.cuda.device(self._device()):
                if train_step.supported(self):         # one autograd node over the C training step (csrc/snv_train.hip)
                    return train_step.run(self, None, distal_input[:, 0:self.in_channels, :])
                mid, large = _train_towers(self, distal_input[:, 0:self.in_channels, :])
                T.flush_bn_ticks()
                return T.Head.apply(None, mid, large)
        return self._forward_dense(None, distal_input[:, 0:self.in_channels, :])


class Network2(_HipSnvBase):
    """Combined local + expanded model (model_snv.py:290-525)."""
    model_no = 2

    def __init__(self, emb_dims, no_of_cont, lin_layer_sizes, emb_dropout, lin_layer_dropouts, in_channels, out_channels,
                 kernel_size, distal_radius, distal_order, distal_fc_dropout, n_class, emb_padding_idx=None):
        super().__init__()
        self.n_class, self.in_channels, self.kernel_size = n_class, in_channels, kernel_size
        self.out_channels = out_channels
        _add_local(self, emb_dims, no_of_cont, lin_layer_sizes, emb_dropout, lin_layer_dropouts, emb_padding_idx)
        self.seq_len = distal_radius * 2 + 1 - (distal_order - 1)
        _add_tower(self, "", in_channels, out_channels, kernel_size, POOLS_MID, distal_fc_dropout, n_class)
        _add_tower(self, "_2", in_channels, out_channels, kernel_size, POOLS_LARGE, distal_fc_dropout, n_class)
        self.local_fc = nn.Sequential(nn.Linear(lin_layer_sizes[-1], n_class))
        self._hip_init()

    def _shape_and_params(self):
        hp = _HostParams()
        params = _lib.MuralSnvParams()
        params.local = hp.local(self, self.local_fc[0])
        params.mid, params.large = hp.tower(self, ""), hp.tower(self, "_2")
        shape = _shape(2, self.n_class, self.no_of_cat, self.emb_layer.num_embeddings, self.lin_layers[0].out_features,
                       self.lin_layers[1].out_features, self.out_channels, self.kernel_size, self.seq_len)
        return shape, params, hp

    def forward(self, local_input, distal_input, _taps=None):
        cont_data, cat_data = local_input
        sym = _symbol_windows(self, distal_input)
        if sym is not None:
            from . import train_step
            cat_data, _ = self._train_inputs(cat_data, None)
            with torch.cuda.device(self._device()):
                return train_step.run(self, cat_data, sym)
        assert distal_input.shape[2] > 200, "Error: distal seq len must be >200"
        if distal_input.shape[2] != self.seq_len:
            raise ValueError(f"distal_input length {distal_input.shape[2]} != 2*distal_radius+1 = {self.seq_len}")
        if self.training:
            from . import train_ops as T
            from . import train_step
            cat_data, distal_input = self._train_inputs(cat_data, distal_input)
            with torch.cuda.device(self._device()):
                if train_step.supported(self):         # one autograd node over the C training step (csrc/snv_train.hip)
                    return train_step.run(self, cat_data, distal_input[:, 0:self.in_channels, :])
                loc = T.local_forward(self, cat_data, self.local_fc[0], self.emb_dropout_layer.p,
                                      [d.p for d in self.droput_layers])
                mid, large = _train_towers(self, distal_input[:, 0:self.in_channels, :])
                T.flush_bn_ticks()
                return T.Head.apply(loc, mid, large)
        return self._forward_dense(cat_data, distal_input[:, 0:self.in_channels, :], taps=_taps)
