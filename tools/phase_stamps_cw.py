"""GPU box: where a launch of the wave-private training conv (forward) spends its time -- per-workgroup wall-clock stamps at entry,
behind the prologue, behind the unit loop and at exit (mural_debug_cw_set_stamps), next to the launch-to-launch time on the host."""
import os
os.environ.setdefault("MURAL_HIP_FLAVOR", "debug")      # validation hooks / development switches: the debug flavour of the library
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mural_amd import _lib  # noqa: E402

lib = _lib.lib()
dev = torch.device("cuda", 0)
B = 4096
for L in (134, 20, 7):
    x = torch.randn(B, L, 32, device=dev)
    y = torch.empty_like(x)
    W = torch.randn(32, 32, 3, device=dev) * 0.1
    bias, gamma, beta = torch.randn(32, device=dev), torch.rand(32, device=dev) + 0.5, torch.randn(32, device=dev)
    state = torch.empty(4, 32, device=dev)
    rm, rv = torch.zeros(32, device=dev), torch.ones(32, device=dev)
    acc = torch.zeros(32 * 2 * 32, dtype=torch.float64, device=dev)
    acc_out = torch.zeros_like(acc)
    st = _lib.current_stream_ptr(dev)
    _lib.check(lib.mural_debug_cl_bn_stats(x.data_ptr(), B * L, 1, acc.data_ptr(), st))
    wfs = torch.empty(6144, device=dev)
    _lib.check(lib.mural_debug_cw_wfrag(W.data_ptr(), wfs.data_ptr(), st))
    stamps = torch.zeros(512 * 4, dtype=torch.int64, device=dev)

    def run():
        _lib.check(lib.mural_debug_cw_conv32_fwd(x.data_ptr(), B, L, 1, acc.data_ptr(), gamma.data_ptr(), beta.data_ptr(), rm.data_ptr(),
                                                 rv.data_ptr(), state.data_ptr(), W.data_ptr(), bias.data_ptr(), 0, None, None,
                                                 acc_out.data_ptr(), 1, y.data_ptr(), wfs.data_ptr(), st))
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        run()
    torch.cuda.synchronize()
    per = (time.perf_counter() - t0) / 50 * 1e6
    _lib.check(lib.mural_debug_cw_set_stamps(stamps.data_ptr()))
    run()
    torch.cuda.synchronize()
    _lib.check(lib.mural_debug_cw_set_stamps(None))
    s = stamps.view(512, 4).cpu().double()
    s = s[s[:, 0] > 0]
    t0g = s[:, 0].min()
    us = (s - t0g) / 100.0
    print("L=%3d: %.1f us launch to launch; %d workgroups; entry %.1f .. %.1f us; prologue done %.1f .. %.1f; units done %.1f .. %.1f; "
          "exit %.1f .. %.1f (mean phase lengths: prologue %.1f, units %.1f, epilogue %.1f)"
          % (L, per, len(s), us[:, 0].min(), us[:, 0].max(), us[:, 1].min(), us[:, 1].max(), us[:, 2].min(), us[:, 2].max(),
             us[:, 3].min(), us[:, 3].max(), (us[:, 1] - us[:, 0]).mean(), (us[:, 2] - us[:, 1]).mean(), (us[:, 3] - us[:, 2]).mean()))
    # who finishes late?  unit-phase end by XCD (workgroup i -> XCD i % 8) and by dispatch order
    import numpy as np
    ids = torch.nonzero(stamps.view(512, 4)[:, 0].cpu() > 0).flatten().numpy()
    end = us[:, 2].numpy()
    if L == 134:
        print("   by XCD      :", " ".join("%.1f" % end[ids % 8 == k].mean() for k in range(8)))
        print("   by id // 64 :", " ".join("%.1f" % end[ids // 64 == k].mean() for k in range(8)))
        print("   start by id // 64:", " ".join("%.1f" % us[:, 0].numpy()[ids // 64 == k].mean() for k in range(8)))
        order = np.argsort(end)
        print("   slowest ids :", ids[order[-16:]].tolist())
        print("   fastest ids :", ids[order[:16]].tolist())
