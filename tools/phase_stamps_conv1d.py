"""Diagnostic: where the time of a short MFMA conv launch goes (mural_debug_conv1d_set_stamps: 5 s_memrealtime values per workgroup --
start, tile staged, MFMAs done, stores issued, stores landed; 100 MHz clock = 10 ns units)."""
import os
os.environ.setdefault("MURAL_HIP_FLAVOR", "debug")      # validation hooks / development switches: the debug flavour of the library
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mural_amd import _lib
lib = _lib.lib()
dev = torch.device("cuda", 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
# (Cin, Cout, Lin, K, stride, up, bias, res)
cases = [(40, 80, 16, 5, 1, 1, 0, 0), (80, 40, 16, 1, 1, 1, 0, 0), (32, 64, 80, 5, 1, 1, 0, 0), (64, 32, 80, 1, 1, 1, 0, 0), (32, 40, 80, 7, 5, 1, 1, 0),
         (48, 96, 8, 5, 1, 1, 0, 0), (24, 48, 400, 5, 1, 1, 0, 0), (48, 24, 400, 1, 1, 1, 0, 1),
         (24, 16, 400, 7, 1, 5, 1, 0), (32, 24, 80, 7, 1, 5, 1, 0)]      # (up > 1: the polyphase form, engine 3)
for Cin, Cout, Lin, K, stride, up, has_bias, nres in cases:
    pad = (K - 1) // 2
    Lout = (Lin * up + 2 * pad - K) // stride + 1
    x = torch.randn(B, Cin, Lin, device=dev)
    wt = torch.randn(Cin, K, Cout, device=dev)
    bias = torch.randn(Cout, device=dev)
    out = torch.empty(B, Cout, Lout, device=dev)
    res = torch.randn(B, Cout, Lout, device=dev)
    st = _lib.current_stream_ptr(dev)
    stamps = torch.zeros(5 * 262144, dtype=torch.int64, device=dev)

    def run():
        _lib.check(lib.mural_debug_conv1d(x.data_ptr(), wt.data_ptr(), bias.data_ptr() if has_bias else None, out.data_ptr(), B, Cin, Lin, Cout, Lout,
                                          K, stride, up, 0, res.data_ptr() if nres else None, None, 3 if up > 1 else 1, st))
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    lib.mural_debug_conv1d_set_stamps(stamps.data_ptr())
    stamps.zero_()
    run()
    torch.cuda.synchronize()
    lib.mural_debug_conv1d_set_stamps(None)
    s = stamps.view(-1, 5).cpu()
    s = s[s[:, 0] > 0]
    t0 = int(s[:, 0].min())
    d = (s - t0).double() * 0.01      # us
    ph = torch.stack([d[:, 0], d[:, 1] - d[:, 0], d[:, 2] - d[:, 1], d[:, 3] - d[:, 2], d[:, 4] - d[:, 3]], 1)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(20):
        run()
    ev1.record()
    torch.cuda.synchronize()
    print("Cin %3d Cout %3d L %4d K %d s %d: %4d WGs; start offset mean %.1f max %.1f | stage %.1f | mfma %.1f | epilogue %.1f | stores land %.1f | last end %.1f us; launch %.1f us"
          % (Cin, Cout, Lin, K, stride, s.shape[0], ph[:, 0].mean(), ph[:, 0].max(), ph[:, 1].mean(), ph[:, 2].mean(), ph[:, 3].mean(), ph[:, 4].mean(),
             d[:, 4].max(), ev0.elapsed_time(ev1) / 20 * 1e3))
