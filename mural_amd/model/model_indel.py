"""Drop-in INDEL model (``UNet_Small``) backed by the gfx950 HIP library.

Mirror of the reference's MuRaL/model/model_indel.py: ``ConvBlock`` (:6-19) and ``UNet_Small`` (:21-176) with the same
constructor signature and sub-module names, hence the same ``state_dict()`` keys (232 keys / 178,036 parameters for the
human insertion model, 225 / 177,912 without the strand-symmetrising ``conv``), so shipped checkpoints load strictly.
The torch sub-modules are parameter containers; ``forward`` runs the fused HIP layer program in eval mode and the
differentiable per-layer HIP ops of ``indel_train.py`` in training mode.  No CPU path.
"""
import ctypes as C

import numpy as np
import torch
import torch.nn as nn

from .. import _lib


class ConvBlock(nn.Module):
    def __init__(self, inp, oup, expand_ratio=2, fused=True):
        super().__init__()
        hidden_dim = round(inp * expand_ratio)
        self.conv = nn.Sequential(
            nn.Conv1d(inp, hidden_dim, 5, 1, padding=2, bias=False), nn.BatchNorm1d(hidden_dim), nn.SiLU(inplace=False),
            nn.Conv1d(hidden_dim, oup, 1, 1, 0, bias=False), nn.BatchNorm1d(oup))


class UNet_Small(nn.Module):
    N_LEVELS = 6

    def __init__(self, n_class, out_channels, kernel_size, downsize, use_reverse=None):
        super().__init__()
        self.n_class, self.out_channels, self.kernel_size = n_class, out_channels, kernel_size
        self.downsize = [int(d) for d in downsize]
        self.use_reverse = use_reverse
        if len(self.downsize) != self.N_LEVELS:
            raise ValueError("down_list must have 6 entries")
        pad = (kernel_size - 1) // 2
        if self.use_reverse:
            self.conv = nn.Sequential(nn.Conv1d(4, 4, kernel_size=kernel_size, padding=pad), nn.BatchNorm1d(4))
        ch = [out_channels * (i + 1) for i in range(self.N_LEVELS)]
        self.channels = ch
        cin = [4] + ch[:-1]
        self.uplblocks = nn.ModuleList([
            nn.Sequential(nn.Conv1d(cin[i], ch[i], stride=self.downsize[i], kernel_size=kernel_size, padding=pad),
                          nn.BatchNorm1d(ch[i])) for i in range(self.N_LEVELS)])
        self.upblocks = nn.ModuleList([nn.Sequential(ConvBlock(c, c, fused=True)) for c in ch])
        self.downlblocks = nn.ModuleList([
            nn.Sequential(nn.Upsample(scale_factor=self.downsize[self.N_LEVELS - 1 - j]),
                          nn.Conv1d(ch[self.N_LEVELS - 1 - j], ch[self.N_LEVELS - 2 - j], kernel_size=kernel_size, padding=pad),
                          nn.BatchNorm1d(ch[self.N_LEVELS - 2 - j])) for j in range(self.N_LEVELS - 1)])
        self.downblocks = nn.ModuleList([nn.Sequential(ConvBlock(ch[self.N_LEVELS - 2 - j], ch[self.N_LEVELS - 2 - j], fused=True))
                                         for j in range(self.N_LEVELS - 1)])
        self.out_conv = nn.Sequential(nn.Conv1d(ch[0], ch[0], kernel_size=1), nn.BatchNorm1d(ch[0]), nn.ReLU(inplace=True),
                                      nn.Conv1d(ch[0], ch[0], kernel_size=1), nn.Softplus())
        self.out_fc = nn.Sequential(nn.BatchNorm1d(ch[0]), nn.Dropout(0.1), nn.Linear(ch[0], n_class), nn.Softplus())
        self._handles = {}
        self._ws = None

    # ------------------------------------------------------------------------------------------------------
    # The folded eval-mode copy is rebuilt when a parameter changed through torch and on every event that can change
    # parameters or BatchNorm buffers behind torch's back (train()/eval() transitions: the training kernels and graph replays
    # update running statistics without bumping tensor versions; load_state_dict; .to()) -- same policy as model_snv.py.
    def invalidate_folded(self):
        self._release()
        self._plist = None
        self._train_layout = None      # tensor objects / storages may have been replaced

    def train(self, mode=True):
        if bool(mode) != self.training:     # a real transition; model.eval() on a model in eval mode keeps the folded copy
            self.invalidate_folded()
        return super().train(mode)

    def _apply(self, fn, *args, **kwargs):
        sig = lambda: tuple((t.data_ptr(), t.dtype, t.device) for t in list(self.parameters()) + list(self.buffers()))   # noqa: E731
        before = sig()
        out = super()._apply(fn, *args, **kwargs)
        if sig() != before:                 # .to() / .cuda() that moved or cast something (a no-op .to(device) does not)
            self.invalidate_folded()
        return out

    def load_state_dict(self, *args, **kwargs):
        self.invalidate_folded()
        return super().load_state_dict(*args, **kwargs)

    def _state_key(self):
        if getattr(self, "_plist", None) is None:
            # parameters and the BatchNorm running statistics (num_batches_tracked does not enter the eval-mode arithmetic)
            self._plist = list(self.parameters()) + [b for b in self.buffers() if b.is_floating_point()]
        return tuple([(t.data_ptr(), t._version) for t in self._plist])

    def _params(self, keep):
        def ptr(t):
            a = np.ascontiguousarray(t.detach().to("cpu", torch.float32).numpy())
            keep.append(a)
            return a.ctypes.data_as(C.c_void_p)

        bn = lambda m: _lib.MuralBN(ptr(m.weight), ptr(m.bias), ptr(m.running_mean), ptr(m.running_var))
        aff = lambda m: _lib.MuralAffine(ptr(m.weight), ptr(m.bias))
        convbn = lambda conv, b: _lib.MuralConvBN(aff(conv), bn(b))
        block = lambda cb: _lib.MuralConvBlock(ptr(cb.conv[0].weight), bn(cb.conv[1]), ptr(cb.conv[3].weight), bn(cb.conv[4]))
        p = _lib.MuralIndelParams()
        if self.use_reverse:
            p.sym = convbn(self.conv[0], self.conv[1])
        for i in range(self.N_LEVELS):
            p.up_l[i] = convbn(self.uplblocks[i][0], self.uplblocks[i][1])
            p.up_b[i] = block(self.upblocks[i][0])
        for j in range(self.N_LEVELS - 1):
            p.down_l[j] = convbn(self.downlblocks[j][1], self.downlblocks[j][2])
            p.down_b[j] = block(self.downblocks[j][0])
        p.out1, p.out_bn, p.out2 = aff(self.out_conv[0]), bn(self.out_conv[1]), aff(self.out_conv[3])
        p.fc_bn, p.fc = bn(self.out_fc[0]), aff(self.out_fc[2])
        return p

    def _get_handle(self, length):
        key = (length, self._state_key())
        if key not in self._handles:
            self._release()
            shape = _lib.MuralIndelShape(self.n_class, self.out_channels, self.kernel_size,
                                         (C.c_int32 * 6)(*self.downsize), int(bool(self.use_reverse)), int(length), 1e-5)
            keep = []
            params = self._params(keep)
            h = C.c_void_p()
            _lib.check(_lib.lib().mural_indel_model_create(C.byref(shape), C.byref(params), C.byref(h)))
            self._handles[key] = h
        return self._handles[key]

    def _release(self):
        for h in getattr(self, "_handles", {}).values():
            _lib.lib().mural_indel_model_destroy(h)
        self._handles = {}

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    def forward(self, distal_input):
        """Forward propagation of a batch: (B, 4, 2*distal_radius) fp32 one-hot -> (B, n_class) Softplus scores."""
        dev = next(self.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("mural_amd models run on a HIP device only: call model.to('cuda') first")
        x = _lib.require_cuda(distal_input, "distal_input").to(torch.float32).contiguous()
        if x.dim() != 3 or x.shape[1] != 4:
            raise ValueError(f"distal_input must be (B, 4, L), got {tuple(x.shape)}")
        if self.training:       # batch-statistics BatchNorm + dropout, differentiable: one C call per direction
            import os
            with torch.cuda.device(dev):
                if os.environ.get("MURAL_INDEL_TRAIN_PER_UNIT"):      # the per-unit autograd composition (model/indel_train.py)
                    from .indel_train import unet_forward_train
                    return unet_forward_train(self, x)
                from . import indel_train_step
                return indel_train_step.run(self, x)
        n, length = x.shape[0], x.shape[2]
        with torch.cuda.device(dev):
            handle = self._get_handle(length)
            out = torch.empty((n, self.n_class), dtype=torch.float32, device=dev)
            need = int(_lib.lib().mural_indel_workspace_bytes(handle, max(n, 1)))
            if self._ws is None or self._ws.numel() < need or self._ws.device != dev:
                self._ws = torch.empty(need, dtype=torch.uint8, device=dev)
            _lib.check(_lib.lib().mural_indel_forward_dense(handle, x.data_ptr(), n, out.data_ptr(), self._ws.data_ptr(),
                                                           self._ws.numel(), _lib.current_stream_ptr(dev)))
        return out

    def forward_packed(self, genome, pos, strand, distal_radius):
        """Scores for sites of a PackedGenome (indel window: [start-R+1, start+R]).  Eval mode: ONE library call
        (``mural_indel_forward_packed``) -- the window is decoded inside the first level's kernel, the one-hot tensor never exists;
        training mode encodes the windows and takes the differentiable path."""
        if self.training:
            return self.forward(genome.encode_onehot(pos, strand, distal_radius, "indel"))
        dev = next(self.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("mural_amd models run on a HIP device only: call model.to('cuda') first")
        pos = _lib.require_cuda(torch.as_tensor(pos, device=dev), "pos").to(torch.int64).contiguous()
        strand = _lib.require_cuda(torch.as_tensor(strand, device=dev), "strand").to(torch.uint8).contiguous()
        if pos.shape != strand.shape or pos.dim() != 1:
            raise ValueError("pos and strand must be 1-D and of equal length")
        n, length = pos.shape[0], 2 * int(distal_radius)
        with torch.cuda.device(dev):
            handle = self._get_handle(length)
            out = torch.empty((n, self.n_class), dtype=torch.float32, device=dev)
            need = int(_lib.lib().mural_indel_workspace_bytes(handle, max(n, 1)))
            if self._ws is None or self._ws.numel() < need or self._ws.device != dev:
                self._ws = torch.empty(need, dtype=torch.uint8, device=dev)
            g = genome.as_struct(dev)
            _lib.check(_lib.lib().mural_indel_forward_packed(handle, C.byref(g), pos.data_ptr(), strand.data_ptr(), n, out.data_ptr(),
                                                            self._ws.data_ptr(), self._ws.numel(), _lib.current_stream_ptr(dev)))
        return out

    def reverse_input(self, distal_input):
        return distal_input.flip([1, 2])
