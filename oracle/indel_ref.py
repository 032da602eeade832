"""CPU oracle for the INDEL model (UNet_Small) -- TEST INFRASTRUCTURE ONLY.

Plain PyTorch fp32 restatement of reference MuRaL/model/model_indel.py:
  * ConvBlock   :6-19   x + BN(Conv1x1(SiLU(BN(Conv5(x)))))   (both convs bias-free)
  * UNet_Small  :21-176 strand-symmetrising conv (:29-32,:154-155), 6 strided encoder levels
                        (:35-84,:157-163), 5 nearest-upsample decoder levels with skip adds
                        (:86-134,:165-170), 1x1 head + Softplus, global max, BN/Dropout/Linear/Softplus
                        (:136-149,:172-174)
Sub-module names equal the reference's so that state_dict keys match.
Pinned by tests/golden/indel_*.npz.  Never imported by the product.
"""
import torch
import torch.nn as nn

N_LEVELS = 6


class ConvBlock(nn.Module):
    def __init__(self, inp, oup, expand_ratio=2):
        super().__init__()
        hid = round(inp * expand_ratio)
        self.conv = nn.Sequential(
            nn.Conv1d(inp, hid, 5, 1, padding=2, bias=False), nn.BatchNorm1d(hid), nn.SiLU(),
            nn.Conv1d(hid, oup, 1, 1, 0, bias=False), nn.BatchNorm1d(oup))

    def forward(self, x):
        return x + self.conv(x)


class UNet_Small(nn.Module):
    def __init__(self, n_class, out_channels, kernel_size, downsize, use_reverse=None):
        super().__init__()
        self.use_reverse = use_reverse
        pad = (kernel_size - 1) // 2
        if use_reverse:
            self.conv = nn.Sequential(nn.Conv1d(4, 4, kernel_size, padding=pad), nn.BatchNorm1d(4))
        ch = [out_channels * (i + 1) for i in range(N_LEVELS)]
        self.channels = ch
        cin = [4] + ch[:-1]
        self.uplblocks = nn.ModuleList([
            nn.Sequential(nn.Conv1d(cin[i], ch[i], kernel_size, stride=downsize[i], padding=pad), nn.BatchNorm1d(ch[i]))
            for i in range(N_LEVELS)])
        self.upblocks = nn.ModuleList([nn.Sequential(ConvBlock(c, c)) for c in ch])
        self.downlblocks = nn.ModuleList([
            nn.Sequential(nn.Upsample(scale_factor=downsize[N_LEVELS - 1 - j]),
                          nn.Conv1d(ch[N_LEVELS - 1 - j], ch[N_LEVELS - 2 - j], kernel_size, padding=pad),
                          nn.BatchNorm1d(ch[N_LEVELS - 2 - j]))
            for j in range(N_LEVELS - 1)])
        self.downblocks = nn.ModuleList([nn.Sequential(ConvBlock(ch[N_LEVELS - 2 - j], ch[N_LEVELS - 2 - j]))
                                         for j in range(N_LEVELS - 1)])
        self.out_conv = nn.Sequential(nn.Conv1d(ch[0], ch[0], 1), nn.BatchNorm1d(ch[0]), nn.ReLU(),
                                      nn.Conv1d(ch[0], ch[0], 1), nn.Softplus())
        self.out_fc = nn.Sequential(nn.BatchNorm1d(ch[0]), nn.Dropout(0.1), nn.Linear(ch[0], n_class), nn.Softplus())

    def forward(self, x, taps=None):
        rec = (lambda n, t: taps.__setitem__(n, t.detach().clone())) if taps is not None else (lambda *_: None)
        h = x
        if self.use_reverse:
            # second term: reverse-complement in (channel+length flip), map back (length flip only)
            h = self.conv(h) + self.conv(h.flip([1, 2])).flip([2])
            rec("sym", h)
        enc = []
        for i in range(N_LEVELS):
            h = self.upblocks[i](self.uplblocks[i](h))
            rec(f"enc{i}", h)
            enc.append(h)
        for j in range(N_LEVELS - 1):
            h = self.downblocks[j](self.downlblocks[j](h))
            h = enc[N_LEVELS - 2 - j] + h
            rec(f"dec{j}", h)
        h = self.out_conv(h)
        h = h.max(dim=2).values
        rec("gmax", h)
        return self.out_fc(h)


def build(*, n_class=8, channels=8, ksize=7, down_list=(1, 4, 5, 5, 5, 2), use_reverse=False):
    return UNet_Small(n_class, channels, ksize, list(down_list), use_reverse)
