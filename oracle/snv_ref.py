"""CPU oracle for the SNV model family (Network0/1/2) -- TEST INFRASTRUCTURE ONLY.

Plain PyTorch fp32 restatement of the reference algorithm, written table-driven
so that the registered sub-module names (and therefore ``state_dict()`` keys and
their order) equal the reference's:

  * local branch   -- reference MuRaL/model/model_snv.py:322-339 (ctor), :451-468 (forward)
  * conv towers    -- :350-430 (ctor), :473-513 (forward)
  * residual block -- :794-812 (pre-activation; modules registered twice)
  * head           -- :515-523 (Network2), :284 (Network1), :93 (Network0: raw logits)

Pinned against the reference by tests/golden/snv_*.npz (see oracle/make_golden.py).
Never imported by the product (mural_amd/).
"""
from collections import OrderedDict

import torch
import torch.nn as nn
import torch.nn.functional as F

MID_HALF = 100  # centre crop half-width of the middle-scale tower (model_snv.py:473)

# (pool kernel, stride, pad) of the three max-pools of each tower
# mid tower: model_snv.py:356,361,371 ; large tower: :399,404,414
POOLS_MID = ((3, 3, 1), (3, 3, 1), (3, 3, 1))
POOLS_LARGE = ((15, 15, 7), (7, 7, 3), (3, 3, 1))


def pool_out_len(length, k, s, p):
    """torch MaxPool1d output length, floor mode."""
    return (length + 2 * p - k) // s + 1


class ResBlock(nn.Module):
    """x + conv2(bn2(relu(conv1(bn1(relu(x)))))); model_snv.py:794-812."""

    def __init__(self, channels, ksize):
        super().__init__()
        pad = (ksize - 1) // 2
        self.bn1 = nn.BatchNorm1d(channels)
        self.conv1 = nn.Conv1d(channels, channels, ksize, 1, pad)
        self.bn2 = nn.BatchNorm1d(channels)
        self.conv2 = nn.Conv1d(channels, channels, ksize, 1, pad)
        # the reference registers the same modules a second time inside `layer`
        self.layer = nn.Sequential(nn.ReLU(), self.bn1, self.conv1, nn.ReLU(), self.bn2, self.conv2)

    def forward(self, x):
        h = self.conv1(self.bn1(F.relu(x)))
        h = self.conv2(self.bn2(F.relu(h)))
        return x[:, :, : h.shape[2]] + h


def _bn_conv(cin, cout, k, relu=False):
    mods = [nn.BatchNorm1d(cin), nn.Conv1d(cin, cout, k, 1, (k - 1) // 2)]
    if relu:
        mods.append(nn.ReLU())
    return nn.Sequential(*mods)


def _register_tower(mod, sfx, in_ch, ch, k, pools, drop, n_class):
    """Register one conv tower on `mod` using the reference's attribute names."""
    setattr(mod, "conv1" + sfx, _bn_conv(in_ch, ch, k))
    setattr(mod, "maxpool1" + sfx, nn.MaxPool1d(*pools[0]))
    setattr(mod, "RBs1" + sfx, nn.Sequential(ResBlock(ch, 3), ResBlock(ch, 3)))
    setattr(mod, "maxpool2" + sfx, nn.MaxPool1d(*pools[1]))
    setattr(mod, "conv2" + sfx, _bn_conv(ch, ch, k))
    setattr(mod, "RBs2" + sfx, nn.Sequential(ResBlock(ch, 3), ResBlock(ch, 3)))
    setattr(mod, "maxpool3" + sfx, nn.MaxPool1d(*pools[2]))
    setattr(mod, "conv3" + sfx, _bn_conv(ch, ch, k, relu=True))
    fc_name = "distal_fc1" if sfx == "" else "distal_fc2"
    setattr(mod, fc_name, nn.Sequential(nn.BatchNorm1d(ch), nn.Dropout(drop), nn.Linear(ch, n_class)))


def _run_tower(mod, sfx, x, taps=None):
    g = lambda n: getattr(mod, n + sfx)
    rec = (lambda name, t: taps.__setitem__(name + sfx, t.detach().clone())) if taps is not None else (lambda *_: None)
    h = g("conv1")(x); rec("conv1", h)
    skip = h = g("maxpool1")(h); rec("pool1", h)
    h = g("RBs1")(h)
    h = h + skip[:, :, : h.shape[2]]; rec("rbs1", h)
    h = g("maxpool2")(h); rec("pool2", h)
    skip = h = g("conv2")(h); rec("conv2", h)
    h = g("RBs2")(h)
    h = h + skip[:, :, : h.shape[2]]; rec("rbs2", h)
    h = g("maxpool3")(h); rec("pool3", h)
    h = g("conv3")(h); rec("conv3", h)
    h = h.max(dim=2).values; rec("gmax", h)
    fc = mod.distal_fc1 if sfx == "" else mod.distal_fc2
    h = fc(h); rec("fc", h)
    return h


def _register_local(mod, emb_dims, no_of_cont, sizes, emb_dropout, dropouts, emb_padding_idx):
    mod.no_of_cat = len(emb_dims)
    mod.emb_layer = nn.Embedding(emb_padding_idx + 1, 5)  # ONE shared table, no padding_idx
    mod.no_of_embs = 5 * len(emb_dims)
    mod.no_of_cont = no_of_cont
    widths = [mod.no_of_embs + no_of_cont] + list(sizes)
    mod.lin_layers = nn.ModuleList([nn.Linear(a, b) for a, b in zip(widths[:-1], widths[1:])])
    mod.first_bn_layer = nn.BatchNorm1d(no_of_cont)
    mod.bn_layers = nn.ModuleList([nn.BatchNorm1d(s) for s in sizes])
    mod.emb_dropout_layer = nn.Dropout(emb_dropout)
    mod.droput_layers = nn.ModuleList([nn.Dropout(p) for p in dropouts])


def _run_local(mod, cont, cat):
    h = torch.cat([mod.emb_layer(cat[:, i]) for i in range(mod.no_of_cat)], dim=1)
    h = mod.emb_dropout_layer(h)
    if mod.no_of_cont != 0:
        h = torch.cat([h, mod.first_bn_layer(cont)], dim=1)
    for lin, drop, bn in zip(mod.lin_layers, mod.droput_layers, mod.bn_layers):
        h = drop(bn(F.relu(lin(h))))  # order is Linear -> ReLU -> BN -> Dropout (model_snv.py:466-468)
    return h


class FeedForwardNN(nn.Module):
    def __init__(self, emb_dims, no_of_cont, lin_layer_sizes, emb_dropout, lin_layer_dropouts, n_class,
                 emb_padding_idx=None):
        super().__init__()
        self.n_class = n_class
        _register_local(self, emb_dims, no_of_cont, lin_layer_sizes, emb_dropout, lin_layer_dropouts, emb_padding_idx)
        self.output_layer = nn.Linear(lin_layer_sizes[-1], n_class)

    def forward(self, cont, cat):
        return self.output_layer(_run_local(self, cont, cat))


class Network0(nn.Module):
    """local-only; returns raw logits (model_snv.py:97-108)."""

    def __init__(self, emb_dims, no_of_cont, lin_layer_sizes, emb_dropout, lin_layer_dropouts, n_class,
                 emb_padding_idx=None):
        super().__init__()
        self.model = FeedForwardNN(emb_dims, no_of_cont, lin_layer_sizes, emb_dropout, lin_layer_dropouts,
                                   n_class, emb_padding_idx)

    def forward(self, local_input, distal_input=None):
        cont, cat = local_input
        return self.model(cont, cat)


def _crop_mid(x):
    c = x.shape[2] // 2
    return x[:, :, c - MID_HALF: c + MID_HALF + 1].detach().clone()


class Network1(nn.Module):
    """expanded-only (model_snv.py:111-287)."""

    def __init__(self, in_channels, out_channels, kernel_size, distal_radius, distal_order, distal_fc_dropout,
                 n_class):
        super().__init__()
        self.n_class, self.in_channels, self.kernel_size = n_class, in_channels, kernel_size
        self.seq_len = distal_radius * 2 + 1 - (distal_order - 1)
        _register_tower(self, "", in_channels, out_channels, kernel_size, POOLS_MID, distal_fc_dropout, n_class)
        _register_tower(self, "_2", in_channels, out_channels, kernel_size, POOLS_LARGE, distal_fc_dropout, n_class)

    def forward(self, local_input, distal_input, taps=None):
        assert distal_input.shape[2] > 200, "Error: distal seq len must be >200bp"
        x = distal_input[:, : self.in_channels]
        mid = _run_tower(self, "", _crop_mid(x), taps)
        large = _run_tower(self, "_2", x, taps)
        p = (F.softmax(mid, dim=1) + F.softmax(large, dim=1)) / 2
        return torch.log(torch.clamp(p, min=1e-9))


class Network2(nn.Module):
    """local + expanded (model_snv.py:290-525)."""

    def __init__(self, emb_dims, no_of_cont, lin_layer_sizes, emb_dropout, lin_layer_dropouts, in_channels,
                 out_channels, kernel_size, distal_radius, distal_order, distal_fc_dropout, n_class,
                 emb_padding_idx=None):
        super().__init__()
        self.n_class, self.in_channels, self.kernel_size = n_class, in_channels, kernel_size
        _register_local(self, emb_dims, no_of_cont, lin_layer_sizes, emb_dropout, lin_layer_dropouts, emb_padding_idx)
        self.seq_len = distal_radius * 2 + 1 - (distal_order - 1)
        _register_tower(self, "", in_channels, out_channels, kernel_size, POOLS_MID, distal_fc_dropout, n_class)
        _register_tower(self, "_2", in_channels, out_channels, kernel_size, POOLS_LARGE, distal_fc_dropout, n_class)
        self.local_fc = nn.Sequential(nn.Linear(lin_layer_sizes[-1], n_class))

    def forward(self, local_input, distal_input, taps=None):
        cont, cat = local_input
        loc = _run_local(self, cont, cat)
        assert distal_input.shape[2] > 200, "Error: distal seq len must be >200"
        x = distal_input[:, : self.in_channels]
        mid = _run_tower(self, "", _crop_mid(x), taps)
        loc = self.local_fc(loc)
        large = _run_tower(self, "_2", x, taps)
        if taps is not None:
            taps["local_logits"] = loc.detach().clone()
        distal = (F.softmax(mid, dim=1) + F.softmax(large, dim=1)) / 2
        p = (F.softmax(loc, dim=1) + distal) / 2
        return torch.log(torch.clamp(p, min=1e-9))


def weights_init(m):
    """Reference initialiser, MuRaL/model/nn_utils.py:14-35 (by class name)."""
    name = m.__class__.__name__
    if "Conv1d" in name or "Conv2d" in name:
        nn.init.xavier_uniform_(m.weight)
        if m.bias is not None:
            nn.init.constant_(m.bias, 0)
    elif "Linear" in name:
        nn.init.kaiming_normal_(m.weight)
        if m.bias is not None:
            nn.init.constant_(m.bias, 0)


REGISTRY = {0: Network0, 1: Network1, 2: Network2}


def build(model_no, *, local_radius=10, local_order=3, distal_radius=1000, hidden=(150, 75), channels=32,
          ksize=3, n_class=4, emb_dropout=0.1, local_dropout=0.1, distal_fc_dropout=0.25, n_cont=0):
    """Construct an oracle model from plain hyper-parameters (what model_choice derives,
    MuRaL/model/nn_utils.py:186-231)."""
    n_cols = 2 * local_radius + 1 - (local_order - 1)
    emb_dims = [(4 ** local_order + 1, 2)] * n_cols
    local_kw = dict(emb_dims=emb_dims, no_of_cont=n_cont, lin_layer_sizes=list(hidden), emb_dropout=emb_dropout,
                    lin_layer_dropouts=[local_dropout, local_dropout], n_class=n_class,
                    emb_padding_idx=4 ** local_order)
    tower_kw = dict(in_channels=4 + n_cont, out_channels=channels, kernel_size=ksize, distal_radius=distal_radius,
                    distal_order=1, distal_fc_dropout=distal_fc_dropout, n_class=n_class)
    if model_no == 0:
        return Network0(**local_kw)
    if model_no == 1:
        return Network1(**tower_kw)
    if model_no == 2:
        kw = dict(local_kw); kw.update(tower_kw)
        return Network2(**kw)
    raise ValueError(f"model_no for snv must be one of [0, 1, 2], got {model_no}")
