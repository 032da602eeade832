"""Time the validation-epoch analytics (mural_amd/evaluation.py) on the GPU next to the CPU restatement (oracle/eval_ref.py:
pandas group-bys, the reference's per-row window loop, numpy Newton) on the host cores.  Usage: python tools/bench_analytics.py [rows]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mural_amd import evaluation as E  # noqa: E402
from oracle import eval_ref  # noqa: E402


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    nc, r = 4, 10
    rng = np.random.default_rng(0)
    codes = rng.integers(0, 4, size=(n, 2 * r + 1)).astype(np.int64)
    ctx = codes[:, r - 1] * 4 + codes[:, r + 1]
    rate = 0.01 + 0.004 * ctx
    p_true = np.stack([1 - rate] + [rate / 3] * 3, axis=1)
    label = (rng.random(n)[:, None] > np.cumsum(p_true, axis=1)).sum(axis=1).clip(0, nc - 1)
    prob = (0.8 * p_true + 0.2 * rng.dirichlet([40, 2, 2, 2], size=n)).astype(np.float32)
    start = np.sort(rng.integers(0, 100_000_000, size=n))
    cid = np.zeros(n, np.int32)
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
    d_codes, d_label, d_prob, d_start, d_cid = d(codes), d(label), d(prob), d(start), d(cid)

    gpu = {
        "kmer 3/5/7": timed(lambda: [E.freq_kmer_comp_multi(d_codes, d_label, d_prob, k, nc) for k in (3, 5, 7)]),
        "regional corr 100k/500k": timed(lambda: [E.corr_calc_sub(d_cid, d_start, d_label, d_prob, w) for w in (100000, 500000)]),
        "regional score": timed(lambda: E.regional_score(d_codes, d_label, d_prob, n, [3, 5], nc)),
        "metrics": timed(lambda: E.calibration_metrics(d_prob, d_label)),
        "fit FullDiri": timed(lambda: E.fit_full_dirichlet(d_prob, d_label), reps=1),
    }
    m = min(n, 200_000)          # bounded CPU samples
    mw = min(n, 20_000)
    t0 = time.perf_counter()
    [eval_ref.freq_kmer_comp_multi(codes[:m], label[:m], prob[:m], k, nc) for k in (3, 5, 7)]
    cpu_kmer = (time.perf_counter() - t0) * n / m
    t0 = time.perf_counter()
    chrom = np.array(["chr1"] * mw)
    [eval_ref.corr_calc_sub(chrom, start[:mw], label[:mw], prob[:mw], w) for w in (100000, 500000)]
    cpu_win = (time.perf_counter() - t0) * n / mw
    t0 = time.perf_counter()
    eval_ref.regional_score(codes[:m], label[:m], prob[:m], m, [3, 5], nc)
    cpu_score = (time.perf_counter() - t0) * n / m
    t0 = time.perf_counter()
    eval_ref.calibration_metrics(prob[:m], label[:m])
    cpu_metrics = (time.perf_counter() - t0) * n / m
    t0 = time.perf_counter()
    eval_ref.fit_full_dirichlet(prob[:m], label[:m])
    cpu_fit = (time.perf_counter() - t0) * n / m
    cpu = {"kmer 3/5/7": cpu_kmer, "regional corr 100k/500k": cpu_win, "regional score": cpu_score, "metrics": cpu_metrics,
           "fit FullDiri": cpu_fit}
    print(f"rows {n}; CPU figures are the numpy/pandas restatement on {m} rows ({mw} for the window loop), scaled linearly")
    for k in gpu:
        print(f"  {k:26s} GPU {gpu[k] * 1e3:9.2f} ms   CPU {cpu[k] * 1e3:11.1f} ms   x{cpu[k] / gpu[k]:.0f}")


if __name__ == "__main__":
    main()
