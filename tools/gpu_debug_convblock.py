"""Validation: the fused ConvBlock launch (mural_debug_convblock) against torch float64 -- plain block, with the k = 7 front (upsampled
or not), skip tensor and tail; the 8-channel block in both forms (0 vector ALU, 1 split: convs on the matrix cores).  TIME=1 adds
timings of the two forms at the bench geometry."""
import os
os.environ.setdefault("MURAL_HIP_FLAVOR", "debug")      # validation hooks / development switches: the debug flavour of the library
import os, sys, time
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mural_amd import _lib

L_ = _lib.lib()
ptr = lambda t: None if t is None else t.data_ptr()


def run(form, B, C, L, front, skip, tail, seed, poison=0, poly=False):
    g = torch.Generator().manual_seed(seed)
    rn = lambda *s: torch.randn(*s, generator=g)
    w5, b5 = rn(2 * C, C, 5) / (5 * C) ** 0.5, rn(2 * C) * 0.3            # torch layouts [out][in][k]
    w1, b1 = rn(C, 2 * C, 1) / (2 * C) ** 0.5, rn(C) * 0.3
    if front and front[1] < 0:      # strided front (stride -front[1], source rows of L * stride columns)
        Cf, st = front[0], -front[1]
        fin = rn(B, Cf, L * st)
        fw, fb = rn(C, Cf, 7) / (7 * Cf) ** 0.5, rn(C) * 0.3
        x = F.conv1d(fin.double(), fw.double(), fb.double(), padding=3, stride=st)
        assert x.shape[2] == L
    elif front:
        Cf, up = front
        fin = rn(B, Cf, L // up)
        fw, fb = rn(C, Cf, 7) / (7 * Cf) ** 0.5, rn(C) * 0.3
        x = F.conv1d(fin.double().repeat_interleave(up, dim=2), fw.double(), fb.double(), padding=3)
    else:
        xin = rn(B, C, L)
        x = xin.double()
    res2 = rn(B, C, L) if skip else None
    hid = F.silu(F.conv1d(x, w5.double(), b5.double(), padding=2))
    want = x + F.conv1d(hid, w1.double(), b1.double())
    if skip:
        want = want + res2.double()
    # workgroups per row: 256 outputs each without a front, 252 with one, 248 for the split form with the polyphase front on the matrix cores
    width = 256 if not front else (248 if (poly and front == (16, 4) and C == 8 and form != 0) else 252)
    tiles = (L + width - 1) // width
    if tail:
        wa, ba, wb, bb = rn(C, C, 1) / C ** 0.5, rn(C) * 0.3, rn(C, C, 1) / C ** 0.5, rn(C) * 0.3
        u = F.softplus(F.conv1d(F.relu(F.conv1d(want, wa.double(), ba.double())), wb.double(), bb.double()))
        want = u.max(dim=2).values
    dev = lambda t: t.cuda().contiguous()
    d_w5 = dev(w5.permute(1, 2, 0))                    # [C][5][2C]
    d_w1 = dev(w1[:, :, 0].t())                        # [2C][C]
    d_b5, d_b1 = dev(b5), dev(b1)
    d_x = None if front else dev(xin)
    d_fin = dev(fin) if front else None
    d_fw = dev(fw.permute(1, 2, 0)) if front else None  # [Cf][7][C]
    d_fb = dev(fb) if front else None
    d_res = dev(res2) if skip else None
    out = None if tail else torch.full((B, C, L), float("nan"), device="cuda")
    tmax = torch.full((B, tiles, C), float("-inf"), device="cuda") if tail else None
    d_pw = None
    if front and poly and front[1] == 4:
        # phase p of output column 4 i + p reads source columns i - 1 + d: the taps k with floor((p - 3 + k) / 4) == d - 1 summed
        pw = torch.zeros(4, front[0], 3, C)
        for ph in range(4):
            for k in range(7):
                d = (ph - 3 + k) // 4 + 1
                pw[ph, :, d, :] += fw[:, :, k].t()
        d_pw = dev(pw)
    if tail:
        d_wa, d_ba, d_wb, d_bb = dev(wa[:, :, 0].t()), dev(ba), dev(wb[:, :, 0].t()), dev(bb)
    else:
        d_wa = d_ba = d_wb = d_bb = None
    call = lambda: L_.mural_debug_convblock(ptr(d_x), ptr(d_w5), ptr(d_b5), ptr(d_w1), ptr(d_b1), ptr(d_res), ptr(out), B, C, L, ptr(d_fin),
                                            ptr(d_fw), ptr(d_fb), front[0] if front else 0, front[1] if front else 1, ptr(d_pw), ptr(d_wa), ptr(d_ba),
                                            ptr(d_wb), ptr(d_bb), ptr(tmax), (form if form >= 0 else 0xff) | poison if poison else form, None)
    rc = call()
    assert rc == 0, L_.mural_last_error()
    torch.cuda.synchronize()
    got = tmax.max(dim=1).values if tail else out
    err = float((got.cpu().double() - want).abs().max() / max(1.0, float(want.abs().max())))
    return err, call


if __name__ == "__main__":
    worst = 0.0
    cases = []
    for C in (8, 16, 24):
        for form in ((0, 1) if C == 8 else (-1,)):
            for (B, L) in ((3, 1000), (2, 252), (5, 37), (2, 8000), (3, 400)):
                for front in (None, (4, 1), (16, 4)):
                    if front and L % front[1]:
                        continue
                    for skip in (False, True):
                        for tail in (False, True):
                            cases.append((form, B, C, L, front, skip, tail, False))
                            if front == (16, 4) and C == 8:
                                cases.append((form, B, C, L, front, skip, tail, True))      # polyphase weights handed over
    if os.environ.get("DEEP"):      # the 32-channel block on short rows (convblock_deep.hip) only
        cases = [(-1, B, 32, L, None, skip, False, False) for (B, L) in ((3, 80), (700, 80), (5, 79), (2, 64), (4, 37), (3, 16), (2, 5), (1, 1), (2000, 80))
                 for skip in (False, True)]
        # ... with the level's strided k = 7 conv as the launch's front (24 x 400 -> 32 x 80 at stride 5 is the shipped geometry)
        cases += [(-1, B, 32, L, (Cf, -st), skip, False, False) for (B, L, Cf, st) in ((3, 80, 24, 5), (600, 80, 24, 5), (5, 64, 24, 4), (2, 80, 16, 5),
                                                                                      (4, 37, 24, 8), (3, 20, 8, 4), (7, 80, 24, 1), (2, 16, 12, 4))
                  for skip in (False, True)]
        # ... and the two deepest levels (40 channels x 16 columns, 48 x 8: a pass takes 5 / 12 rows; ragged last passes)
        cases += [(-1, B, C, L, None, skip, False, False) for (C, L) in ((40, 16), (48, 8)) for B in (1, 3, 5, 7, 12, 13, 29, 700, 2048)
                  for skip in (False, True)]
    for i, c in enumerate(cases):
        err, _ = run(*c[:7], seed=i, poison=0x100, poly=c[7])
        bad = not (err <= 3e-6)
        worst = max(worst, err if err == err else 1.0)
        if bad or os.environ.get("VERBOSE"):
            print(c, "err %.2e" % err, "<-- BAD" if bad else "", flush=True)
    print("cases %d worst %.2e" % (len(cases), worst))
    if os.environ.get("TIME") and os.environ.get("DEEP"):
        for skip in (False, True):
            _, call = run(-1, 2048, 32, 80, None, skip, False, seed=1)
            for _ in range(3):
                call()
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(50):
                call()
            torch.cuda.synchronize()
            print("C 32 L 80 skip", skip, "%.1f us" % ((time.perf_counter() - t) / 50 * 1e6), flush=True)
        _, call = run(-1, 2048, 32, 80, (24, -5), False, False, seed=1)
        for _ in range(3):
            call()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(50):
            call()
        torch.cuda.synchronize()
        print("C 32 L 80 strided front 24 x 400: %.1f us" % ((time.perf_counter() - t) / 50 * 1e6), flush=True)
        for Cc, Lc in ((40, 16), (48, 8)):
            _, call = run(-1, 2048, Cc, Lc, None, True, False, seed=1)
            for _ in range(3):
                call()
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(50):
                call()
            torch.cuda.synchronize()
            print("C", Cc, "L", Lc, "skip True %.1f us" % ((time.perf_counter() - t) / 50 * 1e6), flush=True)
    elif os.environ.get("TIME"):
        for Cc, Lc in ((16, 2000), (24, 400)):
            for skip in (False, True):
                _, call = run(-1, 2048, Cc, Lc, None, skip, False, seed=1)
                for _ in range(3):
                    call()
                torch.cuda.synchronize()
                t = time.perf_counter()
                for _ in range(20):
                    call()
                torch.cuda.synchronize()
                print("C", Cc, "L", Lc, "skip", skip, "%.1f us" % ((time.perf_counter() - t) / 20 * 1e6), flush=True)
        for form in (0, 1):
            for front, skip, tail in (((4, 1), False, False), ((16, 4), True, True), (None, False, False)):
                _, call = run(form, 2048, 8, 8000, front, skip, tail, seed=1, poly=True)
                for _ in range(3):
                    call()
                torch.cuda.synchronize()
                t = time.perf_counter()
                for _ in range(10):
                    call()
                torch.cuda.synchronize()
                print("form", form, "front", front, "skip", skip, "tail", tail, "%.1f us" % ((time.perf_counter() - t) / 10 * 1e6), flush=True)
    sys.exit(0 if worst <= 3e-6 else 1)
