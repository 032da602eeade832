"""GPU box: where the training-mode first-layer kernels (first_train_kernel<SLOT, BWD>, csrc/snv_stage1.hip) spend their time -- wall-clock
stamps of wave 0 of every workgroup (mural_debug_first_set_stamps) at entry / tables ready / window in LDS / window indices built / row
done / every wave done / exit, next to the launch-to-launch time, for the large (pool 15) and the mid (pool 3) tower at batch 4096."""
import os
os.environ.setdefault("MURAL_HIP_FLAVOR", "debug")      # validation hooks / development switches: the debug flavour of the library
import sys
import time

os.environ.setdefault("MURAL_DEBUG_FIRST_CL", "1")      # the channel-last form the composed step runs

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mural_amd import _lib  # noqa: E402
from mural_amd.model import train_ops as T  # noqa: E402

lib = _lib.lib()
dev = torch.device("cuda", 0)
B, Lwin = 4096, 2001
sym = torch.randint(0, 4, (B, Lwin), dtype=torch.uint8, device=dev)
for name, col0, L1, pool in (("large", 0, 2001, (15, 15, 7)), ("mid", 900, 201, (3, 3, 1))):
    pk, ps, pp = pool
    L2 = (L1 + 2 * pp - pk) // ps + 1
    W = torch.randn(32, 4, 3, device=dev) * 0.3
    bias, gamma, beta = torch.randn(32, device=dev) * 0.1, torch.rand(4, device=dev) + 0.5, torch.randn(4, device=dev) * 0.1
    rm, rv = torch.zeros(4, device=dev), torch.ones(4, device=dev)
    tab_f, arg_b, scr_f = T._first_plan(32, pk)
    counts = torch.zeros(16, dtype=torch.int64, device=dev)
    tab = torch.empty(tab_f, device=dev)
    y = torch.empty(B, 32, L2, device=dev)
    arg = torch.empty(B * 32 * L2 * arg_b, dtype=torch.uint8, device=dev)
    dy = torch.randn(B, 32, L2, device=dev)
    scratch = torch.empty(scr_f, device=dev)
    dW, db, dg, dbt = torch.empty_like(W), torch.empty(32, device=dev), torch.empty(4, device=dev), torch.empty(4, device=dev)
    st = _lib.current_stream_ptr(dev)

    def fwd():
        counts.zero_()
        _lib.check(lib.mural_op_first_fwd(sym.data_ptr(), B, Lwin, col0, L1, 32, pk, ps, pp, gamma.data_ptr(), beta.data_ptr(), W.data_ptr(),
                                          bias.data_ptr(), 1e-5, 0.1, rm.data_ptr(), rv.data_ptr(), counts.data_ptr(), tab.data_ptr(),
                                          y.data_ptr(), arg.data_ptr(), st))

    def bwd():
        _lib.check(lib.mural_op_first_bwd(dy.data_ptr(), arg.data_ptr(), sym.data_ptr(), B, Lwin, col0, L1, 32, pk, ps, pp, tab.data_ptr(),
                                          W.data_ptr(), scratch.data_ptr(), dW.data_ptr(), db.data_ptr(), dg.data_ptr(), dbt.data_ptr(), st))
    for which, fn in (("fwd", fwd), ("bwd", bwd)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(30):
            fn()
        torch.cuda.synchronize()
        per = (time.perf_counter() - t0) / 30 * 1e6
        stamps = torch.zeros(256 * 8, dtype=torch.int64, device=dev)
        _lib.check(lib.mural_debug_first_set_stamps(stamps.data_ptr()))
        fn()
        torch.cuda.synchronize()
        _lib.check(lib.mural_debug_first_set_stamps(None))
        s = stamps.view(256, 8).cpu().double()
        s = s[s[:, 0] > 0]
        us = (s - s[:, 0].min()) / 100.0
        names = ["entry", "tables", "window", "kwin", "row", "all waves", "exit"]
        cols = [0, 1, 2, 3, 4, 5, 6] if which == "bwd" else [0, 1, 2, 3, 4, 6]
        print("%s %s: %.1f us per call (all launches of the op); %d workgroups" % (name, which, per, len(s)))
        print("   " + "; ".join("%s %.1f..%.1f (mean %.1f)" % (names[c], us[:, c].min(), us[:, c].max(), us[:, c].mean()) for c in cols))
