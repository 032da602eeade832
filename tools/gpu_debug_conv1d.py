"""GPU box: the generic Conv1d kernels of the INDEL path (vector ALU, conv1d.hip; MFMA implicit GEMM, conv1d_mfma.hip) against
torch in float64 on the CPU, over the layer geometries of UNet_Small (strides 4 / 5 / 2, upsampling 2 / 5 / 4, k = 7 / 5 / 1,
16..96 channels, rows of 8..2000 columns) plus ragged batches.  With MURAL_TEST_VERBOSE: microseconds per launch of both."""
import os
os.environ.setdefault("MURAL_HIP_FLAVOR", "debug")      # validation hooks / development switches: the debug flavour of the library
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mural_amd import _lib  # noqa: E402

lib = _lib.lib()
dev = torch.device("cuda", 0)
verbose = bool(os.environ.get("MURAL_TEST_VERBOSE"))
B0 = 2048 if verbose else 37
# (Cin, Cout, Lin, K, stride, up, act, residuals)
cases = [(8, 16, 8000, 7, 4, 1, 0, 0), (16, 24, 2000, 7, 5, 1, 0, 0), (24, 32, 400, 7, 5, 1, 0, 0), (32, 40, 80, 7, 5, 1, 0, 0),
         (40, 48, 16, 7, 2, 1, 0, 0), (48, 40, 8, 7, 1, 2, 0, 0), (40, 32, 16, 7, 1, 5, 0, 0), (32, 24, 80, 7, 1, 5, 0, 0),
         (24, 16, 400, 7, 1, 5, 0, 0), (32, 64, 80, 5, 1, 1, 2, 0), (64, 32, 80, 1, 1, 1, 0, 2), (48, 96, 8, 5, 1, 1, 2, 0),
         (96, 48, 8, 1, 1, 1, 0, 2), (40, 80, 16, 5, 1, 1, 2, 0), (80, 40, 16, 1, 1, 1, 0, 1), (16, 32, 2000, 5, 1, 1, 2, 0),
         (24, 48, 400, 5, 1, 1, 1, 1), (16, 16, 130, 7, 1, 1, 3, 0), (20, 24, 33, 5, 1, 1, 0, 0), (16, 8, 2000, 7, 1, 4, 0, 1),
         (12, 20, 7, 5, 1, 3, 2, 0),
         # the long-row stride-1 layers of the training step (conv1d_direct.hip): forward and input-gradient shapes, ragged rows
         (4, 4, 8000, 7, 1, 1, 0, 0), (4, 8, 8000, 7, 1, 1, 0, 0), (8, 16, 8000, 5, 1, 1, 0, 0), (16, 8, 8000, 5, 1, 1, 0, 1),
         (16, 32, 2000, 5, 1, 1, 2, 2), (32, 16, 2000, 5, 1, 1, 0, 1), (8, 4, 8000, 7, 1, 1, 0, 0), (8, 12, 1003, 3, 1, 1, 1, 1),
         (16, 20, 517, 7, 1, 1, 3, 0), (32, 7, 530, 5, 1, 1, 0, 2), (4, 32, 600, 7, 1, 1, 0, 0),
         # strided convs on the barrier-free kernel: the encoder's down-convs and ragged variants (row ends inside a segment, rows
         # shorter than a segment's reach, stride 2 / 3 / 8, 5 taps, residuals)
         (8, 16, 8003, 7, 4, 1, 1, 0), (16, 24, 1999, 7, 5, 1, 2, 1), (8, 16, 333, 5, 2, 1, 0, 2), (4, 8, 1000, 7, 3, 1, 3, 0),
         (8, 16, 700, 7, 8, 1, 0, 0), (16, 16, 90, 7, 5, 1, 0, 0),
         # up-convs on the polyphase barrier-free kernel: ragged source rows, activation, 80 and 120 GEMM rows, fewer rows than a slice
         (24, 16, 403, 7, 1, 5, 2, 0), (32, 24, 77, 7, 1, 5, 0, 0), (24, 12, 131, 7, 1, 5, 1, 0), (32, 16, 100, 7, 1, 4, 3, 0)]
worst = 0.0
for Cin, Cout, Lin, K, stride, up, act, nres in cases:
    B = (B0 if Lin <= 2000 or not verbose else 512) if not (verbose and Lin >= 2000 and Cin <= 32 and stride == 1 and up == 1) else 128
    pad = (K - 1) // 2
    Lout = (Lin * up + 2 * pad - K) // stride + 1
    g = torch.Generator().manual_seed(Cin * 1000 + Cout + Lin)
    x = torch.randn(B, Cin, Lin, generator=g, dtype=torch.float64)
    W = torch.randn(Cout, Cin, K, generator=g, dtype=torch.float64) / (Cin * K) ** 0.5
    bias = torch.randn(Cout, generator=g, dtype=torch.float64)
    res = [torch.randn(B, Cout, Lout, generator=g, dtype=torch.float64) for _ in range(nres)]
    xin = x.repeat_interleave(up, dim=2) if up > 1 else x
    y = F.conv1d(xin, W, bias, stride=stride, padding=pad)
    y = [y, F.relu(y), F.silu(y), F.softplus(y)][act]
    for r in res:
        y = y + r
    f32 = lambda t: t.to(torch.float32).contiguous().to(dev)
    xd, bd = f32(x), f32(bias)
    wt = f32(W.permute(1, 2, 0))                      # [Cin][K][Cout]
    rd = [f32(r) for r in res] + [None, None]
    st = _lib.current_stream_ptr(dev)
    line = f"Cin {Cin:3d} Cout {Cout:3d} Lin {Lin:5d} K {K} s {stride} up {up} act {act} res {nres}:"
    direct = up == 1 and Cin in (4, 8, 16, 32) and Cout <= 32 and not (Cout > 16 and Cin > 16) and K in (3, 5, 7)
    for engine, name in ((0, "valu"), (1, "mfma")) + (((3, "polyphase"), (5, "direct-poly")) if up > 1 else ()) + (((4, "direct"),) if direct else ()):
        out = torch.full((B, Cout, Lout), float("nan"), device=dev)

        def run(eng=engine):
            _lib.check(lib.mural_debug_conv1d(xd.data_ptr(), wt.data_ptr(), bd.data_ptr(), out.data_ptr(), B, Cin, Lin, Cout, Lout, K, stride,
                                              up, act, None if rd[0] is None else rd[0].data_ptr(),
                                              None if rd[1] is None else rd[1].data_ptr(), eng, st))
        try:
            run(engine | 0x100)                   # every CU's LDS filled with NaN first: no engine may depend on LDS it has not written
        except (RuntimeError, ValueError):        # geometry outside the engine (fewer than 16 GEMM rows)
            line += f"  {name} n/a"
            continue
        err = float((out.double().cpu() - y).abs().max() / (y.abs().max() + 1e-12))
        worst = max(worst, err if err == err else 1.0)
        line += f"  {name} {err:.1e}"
        if verbose:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                run()
            torch.cuda.synchronize()
            line += " %6.1f us" % ((time.perf_counter() - t0) / 20 * 1e6)
    print(line)
print("worst %.2e" % worst)
sys.exit(0 if worst <= 2e-6 else 1)
