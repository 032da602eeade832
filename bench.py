#!/usr/bin/env python3
"""Headline benchmark: predicted bases/s for the default SNV model (local_radius=10, distal_radius=1000, 4-class).

Contract (driver): ``python bench.py --gpus N --steps K --warmup W`` prints ONE JSON line on rank 0.  For N>1 it is
launched by ``python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`` (one rank per GPU, RCCL).

Workload (BASELINE.json configs[1]): a synthetic chromosome of 10,000,000 + 2*1000 i.i.d.
uniform ACGT bases (numpy default_rng(20251121)), stored 2-bit packed and resident in HBM when the timed region
starts; sites = consecutive bases from 1000 on, '+' strand for even / '-' for odd index; weights = S-config
``weights_init`` with torch.manual_seed(0), eval mode.  One *step* = the hot path (k-mer encode + local MLP + fused
window decode + first-layer lookup kernel / conv-tower kernel with head) over one batch of ``--batch`` sites per rank; with the default flags 20 steps x
500,000 sites = the 10M positions of the config.  Every site's full +-1 kb window is evaluated independently -- no
cross-position reuse.  N>1: ranks take disjoint site shards (weak scaling) and each step ends with one RCCL
all_gather of the (batch, 4) fp32 log-probabilities.

Extra objects on the JSON line (N=1, measured after the timed prediction region):
``roofline`` (dominant kernel = snv_tower_wave, the wave-private tower kernel: algorithmic FLOP per launch of the layers it evaluates / HIP-event duration of
that kernel measured live in the timed region, vs the 157.3 TFLOP/s fp32 MFMA peak); ``cpu_baseline`` (oracle = PyTorch-CPU
restatement of the reference on this box's host cores, os.cpu_count() threads, batch 16 / 256 / 1024, bounded samples);
``variants`` (independent random windows handed over as the reference's dense tensors; the reference's default 16-site calls;
``model_predict_m`` over a 16-row loader); ``dense_reuse`` (the SAME 10M-site workload through the cross-position reuse path,
SURVEY.md section 8f-4 -- reported beside the per-window headline, never in place of it); ``train`` (configs[2]: S-config
training step at batch 4096, >= 200 steps, un-synchronised and per-step synchronised, with its roofline) and ``indel``
(configs[3]: UNet_Small with the shipped human insertion weights when the fixture is present, 1e5 positions, with its roofline).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

LOCAL_RADIUS, LOCAL_ORDER, DISTAL_RADIUS, N_CLASS = 10, 3, 1000, 4
GENOME_SITES = 10_000_000
# algorithmic forward FLOP per position, SURVEY.md section 8d (2*Cin*Cout*K*Lout per conv, 2*in*out per linear)
FLOP_TOTAL = 8_096_144                     # local MLP 51,600 + mid tower 2,556,928 + large tower 5,487,616
FLOP_FIRST_LAYERS = 154_368 + 1_536_768    # BN(4)+Conv(4->32): evaluated as 3-mer table lookups by snv_stage1_kernel
FLOP_TOWERS = 2_556_928 + 5_487_616 - FLOP_FIRST_LAYERS   # the 32->32 convs + fc: work of the dominant kernel
PEAK_FP32_MFMA_TFLOPS = 157.3              # /opt/skills/guides/MI355X_MICROARCH.md


def snv_flop_per_site(distal_radius, local_radius=LOCAL_RADIUS, local_order=LOCAL_ORDER, C=32, K=3, n_class=N_CLASS, h1=150, h2=75):
    """Algorithmic forward FLOP per position of Network2 by SURVEY.md section 8d's rule (2*Cin*Cout*K*Lout per Conv1d, 2*in*out per
    Linear, nothing for BatchNorm / ReLU / pools): 8,096,144 at the benchmark's 10 / 1000."""
    pool = lambda L, k, s, p: (L + 2 * p - k) // s + 1      # noqa: E731
    ncol = 2 * local_radius + 1 - (local_order - 1)
    total = 2 * (5 * ncol * h1 + h1 * h2 + h2 * n_class)
    for L1, pools in ((201, ((3, 3, 1),) * 3), (2 * distal_radius + 1, ((15, 15, 7), (7, 7, 3), (3, 3, 1)))):
        L2, L3, L4 = pool(L1, *pools[0]), 0, 0
        L3 = pool(L2, *pools[1])
        L4 = pool(L3, *pools[2])
        total += 2 * 4 * C * K * L1 + 2 * C * C * K * (4 * L2 + 5 * L3 + L4) + 2 * C * n_class
    return total


def build_model(device, distal_radius=DISTAL_RADIUS):
    from mural_amd.model import model_choice, weights_init
    ncol = 2 * LOCAL_RADIUS + 1 - (LOCAL_ORDER - 1)
    cfg = dict(local_radius=LOCAL_RADIUS, local_order=LOCAL_ORDER, local_hidden1_size=150, local_hidden2_size=75,
               distal_radius=distal_radius, emb_dropout=0.1, local_dropout=0.1, CNN_kernel_size=3, CNN_out_channels=32,
               distal_fc_dropout=0.25)
    common = dict(emb_dims=[(4 ** LOCAL_ORDER + 1, 2)] * ncol, n_cont=0, n_class=N_CLASS, distal_order=1, in_channels=4)
    torch.manual_seed(0)
    model = model_choice(2, cfg, common, "snv")
    model.apply(weights_init)
    return model.to(device).eval()


def synthetic_genome(n_bases):
    rng = np.random.default_rng(20251121)
    return rng.integers(0, 4, size=n_bases, dtype=np.uint8)


def pack2(codes):
    n = len(codes)
    two = np.concatenate([codes.astype(np.uint32), np.zeros((-n) % 16, np.uint32)]).reshape(-1, 16)
    packed = np.bitwise_or.reduce(two << (2 * np.arange(16, dtype=np.uint32))[None, :], axis=1).astype(np.uint32)
    mask = np.zeros((n + 31) // 32, np.uint32)
    return packed, mask


def cpu_model_name():
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(model_state, codes, budget_s=8.0):
    """Time the oracle (CPU restatement of the reference, oracle/snv_ref.py) on bounded samples of the workload: model only,
    inputs pre-encoded to the reference's tensor layout, batch 16 (the reference's default, commands/predict.py:90) / 256 / 1024,
    >= 5 warm-up iterations discarded, median of up to 20 timed ones, at the best torch thread count of a scan up to
    os.cpu_count() (three warm iterations, then the median of five per setting; see below)."""
    from oracle import encode_ref, snv_ref
    orc = snv_ref.build(2, local_radius=LOCAL_RADIUS, local_order=LOCAL_ORDER, distal_radius=DISTAL_RADIUS)
    orc.load_state_dict(model_state)
    orc.eval()
    ncpu = os.cpu_count() or 1

    def batch_inputs(it, batch):
        pos = DISTAL_RADIUS + (it * batch) % 100_000 + np.arange(batch)
        sym = ["-" if (p - DISTAL_RADIUS) % 2 else "+" for p in pos]
        return (torch.from_numpy(encode_ref.kmer_encode(codes, pos, sym, LOCAL_RADIUS, LOCAL_ORDER)),
                torch.from_numpy(encode_ref.onehot_encode(codes, pos, sym, DISTAL_RADIUS)))

    # Thread count: SURVEY.md section 8d names os.cpu_count(); on a many-core host intra-op threading of these small convs
    # collapses long before that (measured on the 256-CPU MI355X host: 13 bases/s at 256 threads against several thousand at
    # 16-64), so the scan walks up from 8 threads, stops once a setting is 2x slower than the best (it only gets worse from
    # there) and the baseline is timed at the best setting -- the fair comparison -- with the whole scan reported.
    scan = {}
    with torch.no_grad():
        cont = torch.zeros(256, 1, dtype=torch.float64)
        cat, x = batch_inputs(0, 256)
        for threads in sorted({min(ncpu, t) for t in (8, 16, 32, 64, 128, ncpu)}):
            torch.set_num_threads(threads)
            for _ in range(3):                      # warm: thread pool spin-up, allocator, first-touch of the conv buffers
                orc((cont, cat), x)
            reps = []
            for _ in range(5):
                t0 = time.perf_counter()
                orc((cont, cat), x)
                reps.append(time.perf_counter() - t0)
                if reps[-1] > 2.0:                  # a setting this slow (256 rows in > 2 s) is not the best one: do not wait for 5
                    break
            scan[str(threads)] = 256 / float(np.median(reps))
            if scan[str(threads)] < 0.5 * max(scan.values()):
                break
    best_threads = int(max(scan, key=scan.get))
    torch.set_num_threads(best_threads)

    by_batch, samples, repeats_256 = {}, {}, []
    with torch.no_grad():
        for batch in (16, 256, 1024):
            cont = torch.zeros(batch, 1, dtype=torch.float64)
            meds = []
            for rep in range(3 if batch == 256 else 1):      # the headline batch: three repeats of the loop, median of their medians
                times, used, it = [], 0.0, 0
                while it < 25 and (it < 7 or used < budget_s / (3 if batch == 256 else 1)):
                    cat, x = batch_inputs(it + 25 * rep, batch)
                    t0 = time.perf_counter()
                    orc((cont, cat), x)
                    dt = time.perf_counter() - t0
                    used += dt
                    if it >= (5 if rep == 0 else 1):
                        times.append(dt)
                    it += 1
                meds.append(batch / float(np.median(times)))
            by_batch[str(batch)] = float(np.median(meds))
            if batch == 256:
                repeats_256 = meds
            samples[str(batch)] = f"{len(meds)} x up to {len(times)} timed iterations after warm-up"
    # the training step of the same restatement: CE(sum) + clip + Adam at batch 128 (the reference's default) and 4096
    orc.train()
    opt = torch.optim.Adam(orc.parameters(), lr=1e-3)
    crit = torch.nn.CrossEntropyLoss(reduction="sum")
    train = {}
    for tb, reps in ((128, 3), (4096, 1)):
        cat, x = batch_inputs(0, tb)
        y = torch.from_numpy(np.arange(tb) % N_CLASS)
        cont = torch.zeros(tb, 1, dtype=torch.float64)

        def train_step():
            t0 = time.perf_counter()
            loss = crit(orc((cont, cat), x), y)
            opt.zero_grad()
            loss.backward()
            torch.nn.utils.clip_grad_norm_(orc.parameters(), max_norm=10)
            opt.step()
            return time.perf_counter() - t0

        if tb == 128:
            train_step()
        t = min(train_step() for _ in range(reps))
        train[str(tb)] = {"steps_per_s": 1.0 / t, "sites_per_s": tb / t}
    # SURVEY.md section 8d's second figure: the same batch with a vectorised numpy encode of its inputs inside the timed region (the
    # oracle's encoders: k-mer ids + one-hot windows from base codes; the reference's own per-character Python encoders are ~100 x slower)
    enc_times = []
    for it in range(6):
        t0 = time.perf_counter()
        batch_inputs(100 + it, 256)
        enc_times.append(time.perf_counter() - t0)
    t_encode = float(np.median(enc_times[1:]))
    with_encode = 256.0 / (256.0 / by_batch["256"] + t_encode)
    # batch 256 at the chosen thread count: the median of the medians of three repeats of the timed loop (on this shared host a single
    # loop can catch a slow spell: 4.8 k vs 8.1 k bases/s were seen in one run); the repeats are reported
    value = by_batch["256"]
    return {"value": value, "unit": "bases/s", "cores": best_threads, "kind": "port", "cpu_model": cpu_model_name(),
            "host_cpus": ncpu, "threads_scan_bases_per_s_at_batch_256": scan,
            "sample": "model only (inputs pre-encoded: cat_x int64, distal_x fp32 one-hot) on windows of the same synthetic chromosome; "
                      "value = batch 256 (median of the medians of three repeats of the timed loop); " + "; ".join(f"batch {b}: {v}" for b, v in samples.items()),
            "bases_per_s_by_batch": by_batch, "batch_256_repeat_medians": repeats_256, "train": train,
            "bases_per_s_with_encode": with_encode, "encode_seconds_per_256": t_encode,
            "with_encode_note": "batch 256: model time + a vectorised numpy encode of the batch (k-mer ids, fp32 one-hot windows) from base codes",
            "train_note": "forward + backward + clip + Adam of the same restatement; batch 4096 is a single timed step"}


PEAK_HBM_TBS = 8.0                         # /opt/skills/guides/MI355X_MICROARCH.md
FLOP_TRAIN_PER_SITE = 22.60e6              # fwd + dgrad + wgrad, no dgrad for the two input convs
FLOP_INDEL_PER_POS = 113.4e6               # UNet_Small insertion geometry, L = 8000


def profile_fact(name):
    """Numbers that need a profiler pass (PMC HBM bytes, launches per step) are read from the committed summary of that pass
    (profiles/<name>.json, written from a rocprofv3 run of this very command on an MI355X); the JSON carries its provenance."""
    path = os.path.join(ROOT, "profiles", name + ".json")
    if not os.path.exists(path):
        return None
    with open(path) as fh:
        return json.load(fh)


def _freeze_host_heap():
    from mural_amd.train import freeze_host_heap
    freeze_host_heap()      # what mural_amd.train.train_epoch does before its loop: keeps full GC passes out of the step


def train_steps_per_s(device, genome, B=4096, steps=1000, warmup=20, sync_steps=50):
    """BASELINE.json configs[2]: S-config from scratch, batch 4096, Adam lr 1e-3, CE-sum, clip 10, default dropouts.  `steps`
    steps without a host synchronisation in between (windows encoded from the packed genome inside the timed loop), then
    `sync_steps` individually synchronised ones (the reference reads loss.item() every step, training.py:437)."""
    import torch.nn as nn
    from mural_amd.train import Adam, CrossEntropySum, clip_grad_norm_
    model = build_model(device).train()
    # mural_amd.train.Adam = torch.optim.Adam with the update as ONE launch over the flat parameter / gradient / moment buffers of the
    # library's training step (same rule; tests/test_gpu_train.py compares the two); MURAL_BENCH_TORCH_ADAM=1: torch's fused multi-tensor step
    opt = (torch.optim.Adam(model.parameters(), lr=1e-3, fused=True) if os.environ.get("MURAL_BENCH_TORCH_ADAM") == "1"
           else Adam(model.parameters(), lr=1e-3))
    crit = CrossEntropySum()                  # nn.CrossEntropyLoss(reduction="sum") in one launch per direction (mural_amd.train)
    rng = np.random.default_rng(1)
    total = steps + warmup + sync_steps
    labels = torch.from_numpy(rng.choice(4, size=total * B, p=[0.955, 0.015, 0.015, 0.015]).astype(np.int64)).to(device)
    cont = torch.zeros(B, 1, device=device)

    # the site list of every step (like the labels above: inputs resident in HBM); the windows and k-mer columns are encoded from the
    # packed genome inside the loop
    idx_all = torch.arange(total * B, device=device) % GENOME_SITES
    pos_all, strand_all = idx_all + DISTAL_RADIUS, (idx_all & 1).to(torch.uint8)
    del idx_all

    def step(s, dense=False):
        pos, strand = pos_all[s * B:(s + 1) * B], strand_all[s * B:(s + 1) * B]
        cat = genome.encode_kmer(pos, strand, LOCAL_RADIUS, LOCAL_ORDER)
        # one symbol per column (the training step's own input form) or, for the side figure, the reference loader's one-hot tensor
        x = genome.encode_onehot(pos, strand, DISTAL_RADIUS) if dense else genome.encode_symbols(pos, strand, DISTAL_RADIUS)
        loss = crit(model((cont, cat), x), labels[s * B:(s + 1) * B])
        opt.zero_grad()
        loss.backward()
        clip_grad_norm_(model, 10)            # mural_amd.train: torch.nn.utils.clip_grad_norm_ over the flat gradient buffer
        opt.step()
        return loss

    for s in range(warmup):
        step(s)
    _freeze_host_heap()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(warmup, warmup + steps):
        loss = step(s)
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / steps
    times = []
    for s in range(warmup + steps, total):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        step(s).item()
        times.append(time.perf_counter() - t1)
    t_sync = float(np.median(times))
    # side figure: the same loop fed with dense one-hot windows (16 bytes per column written by the encoder and read back by the step)
    dense_steps = min(200, steps)
    for s in range(min(20, steps)):      # (the 131 MB window tensors of this route settle in the caching allocator first: with three warm-up
        step(s, dense=True)              # steps the timed loop still contained allocations, 385 vs 500-520 steps/s in its first 200 steps)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    for s in range(dense_steps):
        step(s, dense=True)
    torch.cuda.synchronize()
    t_dense = (time.perf_counter() - t2) / dense_steps
    tflops = FLOP_TRAIN_PER_SITE * B / t / 1e12
    out = {"steps_per_s": 1.0 / t, "ms_per_step": t * 1e3, "batch": B, "sites_per_s": B / t, "steps": steps,
           "steps_per_s_dense_input": 1.0 / t_dense, "dense_input_steps": dense_steps,
           "steps_per_s_synchronised": 1.0 / t_sync, "ms_per_step_synchronised": t_sync * 1e3, "synchronised_steps": sync_steps,
           "final_loss_per_site": float(loss.item()) / B, "optimizer": "Adam lr 1e-3", "loss": "CrossEntropy(sum), clip_grad_norm 10",
           "note": "forward (batch-statistics BatchNorm, dropout 0.1/0.1/0.25) + backward + clip + Adam; windows encoded from the packed "
                   "genome inside the timed loop as symbol windows (PackedGenome.encode_symbols: bit-identical to the dense one-hot "
                   "route, tests/test_gpu_train.py); steps_per_s = %d steps without a host sync in between" % steps,
           "roofline": {"flop_per_step": FLOP_TRAIN_PER_SITE * B, "achieved_TFLOPs": tflops, "peak_TFLOPs": PEAK_FP32_MFMA_TFLOPS,
                        "frac_mfma": tflops / PEAK_FP32_MFMA_TFLOPS}}
    fact = profile_fact("r06_train_step") or profile_fact("r05_train_step")      # profiled on the symbol route: the loop whose rate steps_per_s is
    if fact:
        hbm = fact.get("hbm_bytes_per_unit", fact.get("hbm_bytes_per_step"))
        gbs = hbm / t / 1e12
        out["roofline"].update({"bound": "hbm", "hbm_bytes_per_step": hbm, "achieved_TBs": gbs, "peak_TBs": PEAK_HBM_TBS,
                                "frac_hbm": gbs / PEAK_HBM_TBS, "launches_per_step": fact.get("launches_per_step"),
                                "traffic_source": fact.get("source")})
    return out


def indel_model_and_weights(device):
    """UNet_Small in the human-insertion configuration; the shipped Homo_sapiens/INDEL/insertion weights when the parity fixture
    that carries them is present (tests/golden travels with the repository), weights_init otherwise."""
    from mural_amd.model import model_choice, weights_init
    cfg = dict(CNN_out_channels=8, CNN_kernel_size=7, down_list=[1, 4, 5, 5, 5, 2], use_reverse=True)
    torch.manual_seed(0)
    model = model_choice(0, cfg, dict(n_class=8), "indel")
    fixture = os.path.join(ROOT, "tests", "golden", "indel_pretrained_human_insertion.npz")
    if os.path.exists(fixture):
        fx = np.load(fixture)
        model.load_state_dict({k[3:]: torch.from_numpy(fx[k]) for k in fx.files if k.startswith("w::")})
        weights = "Homo_sapiens/INDEL/insertion (shipped checkpoint, via tests/golden/indel_pretrained_human_insertion.npz)"
    else:
        model.apply(weights_init)
        weights = "weights_init, torch.manual_seed(0)"
    return model.to(device), weights


def indel_positions_per_s(device, genome, n=196_608, chunk=24_576):
    """BASELINE.json configs[3]: UNet_Small, human-insertion geometry (L=8000, 8 classes, use_reverse), 2e5 positions decoded from
    the packed genome inside the timed region (8 calls of 24576 positions = 6 internal chunks of 4096 each, two in flight; two
    warm-up calls)."""
    model, weights = indel_model_and_weights(device)
    model.eval()
    idx = torch.arange(n, device=device, dtype=torch.int64)
    pos, strand = idx * 47 + 4000, (idx & 1).to(torch.uint8)

    def run():
        for c0 in range(0, n, chunk):
            model.forward_packed(genome, pos[c0:c0 + chunk], strand[c0:c0 + chunk], 4000)

    with torch.no_grad():
        for _ in range(2):
            model.forward_packed(genome, pos[:chunk], strand[:chunk], 4000)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    tflops = n / dt * FLOP_INDEL_PER_POS / 1e12
    out = {"positions_per_s": n / dt, "positions": n, "window": 8000, "n_class": 8, "weights": weights,
           "note": "window decode from the packed genome inside the timed region; 113.4 MFLOP/position",
           "roofline": {"bound": "mfma", "achieved": tflops, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                        "frac": tflops / PEAK_FP32_MFMA_TFLOPS}}
    fact = profile_fact("r06_indel_forward_pmc") or profile_fact("r05_indel_forward_pmc")
    if fact:
        per_pos = fact["hbm_bytes_per_unit"] / 2048.0      # (the profiled unit is one forward of 2048 positions)
        tbs = per_pos * n / dt / 1e12
        out["roofline"].update({"hbm_bytes_per_position": per_pos, "achieved_TBs": tbs, "frac_hbm": tbs / PEAK_HBM_TBS,
                                "traffic_source": fact.get("source")})
    # the same model on DENSE one-hot windows (B, 4, 8000) resident in HBM -- what the reference's own loader hands to the model
    # (the drop-in input of an unchanged predict loop): 8 calls of 4096 windows encoded outside the timed region
    nd, cd = 32_768, 4_096
    with torch.no_grad():
        xd = genome.encode_onehot(pos[:cd], strand[:cd], 4000, "indel")
        for _ in range(2):
            model(xd)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(nd // cd):
            model(xd)
        torch.cuda.synchronize()
        dtd = time.perf_counter() - t0
    out["positions_per_s_dense_input"] = nd / dtd
    out["dense_input_note"] = ("dense (B, 4, 8000) one-hot windows resident in HBM, 8 calls of 4096: classified into symbol bytes and taken by the "
                               "same persistent first level as the packed entry (columns that are no symbol are evaluated from their floats)")
    del xd
    # one training configuration of the same model: batch 128 (the reference's default), CE(sum) + clip + Adam.  The step is ~500
    # small launches behind Python autograd glue: the eager loop runs at the speed of the host's Python (7-11 ms on this pool's
    # boxes), mural_amd.train.GraphedIndelTrainStep replays the same step as one HIP graph and is bound by the device alone.
    from mural_amd.train import CrossEntropySum, GraphedIndelTrainStep, clip_grad_norm_
    tb = 128
    model.train()
    crit = CrossEntropySum()                  # nn.CrossEntropyLoss(reduction="sum") in one launch per direction
    x = genome.encode_onehot(pos[:tb], strand[:tb], 4000, "indel")
    y = (idx[:tb] % 8)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, fused=True)       # one multi-tensor launch; same update rule

    def step():
        loss = crit(model(x), y)
        opt.zero_grad()
        loss.backward()
        clip_grad_norm_(model, 10)            # mural_amd.train: the launches of torch.nn.utils.clip_grad_norm_ over a cached parameter list
        opt.step()

    for _ in range(3):
        step()
    _freeze_host_heap()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        step()
    torch.cuda.synchronize()
    dt_eager = (time.perf_counter() - t0) / 20
    dt_graph, graph_error = float("inf"), None
    try:
        opt_g = torch.optim.Adam(model.parameters(), lr=1e-3, capturable=True, fused=True)
        gstep = GraphedIndelTrainStep(model, opt_g, crit, x, y)
        for _ in range(3):
            gstep(x, y)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(40):
            gstep(x, y)
        gstep.finish()
        dt_graph = (time.perf_counter() - t0) / 40
    except Exception as e:      # noqa: BLE001  (the eager number stands on its own)
        graph_error = f"{type(e).__name__}: {e}"[:300]
    dt = dt_eager                               # the loop a user of training.py gets; the graph replay of the same step stands beside it
    tf = 3 * FLOP_INDEL_PER_POS * tb / dt / 1e12
    out["train"] = {"steps_per_s": 1.0 / dt, "ms_per_step": dt * 1e3, "mode": "eager loop",
                    "ms_per_step_eager": dt_eager * 1e3, "ms_per_step_graph_replay": None if graph_error else dt_graph * 1e3,
                    "graph_replay_error": graph_error, "batch": tb,
                    "positions_per_s": tb / dt,
                    "note": "forward (batch-statistics BatchNorm, dropout) + backward + clip + Adam on pre-encoded windows; eager = 20 steps "
                            "of the plain Python loop (one C call per direction, mural_indel_train_forward / _backward: ~320 launches incl. torch's loss / clip / Adam, device-bound), "
                            "graph replay = 40 replays of the same step as one HIP graph (GraphedIndelTrainStep, inputs copied in per step: "
                            "bound by the device alone); ms_per_step is the EAGER loop's",
                    "roofline": {"bound": "mfma", "achieved": tf, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": tf / PEAK_FP32_MFMA_TFLOPS,
                                 "flop_per_step": 3 * FLOP_INDEL_PER_POS * tb}}
    tfact = profile_fact("r06_indel_train") or profile_fact("r05_indel_train")
    if tfact and tfact.get("hbm_bytes_per_step"):
        tb_s = tfact["hbm_bytes_per_step"] / dt / 1e12
        out["train"]["roofline"].update({"hbm_bytes_per_step": tfact["hbm_bytes_per_step"], "achieved_TBs": tb_s, "frac_hbm": tb_s / PEAK_HBM_TBS,
                                         "launches_per_step": tfact.get("launches_per_step"), "traffic_source": tfact.get("source")})
    return out


def workload_variants(device, model, genome):
    """The other call patterns SURVEY.md section 8d asks for, next to the headline."""
    import torch.nn as nn
    from mural_amd.model import model_predict_m
    out = {}

    def timed(fn, reps, warm):
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps

    with torch.no_grad():
        # independent random windows handed over as the reference's own tensors (no shared genome buffer)
        B = 16384
        g = torch.Generator(device=device).manual_seed(5)
        codes = torch.randint(0, 4, (B, 2 * DISTAL_RADIUS + 1), device=device, generator=g)
        x = torch.nn.functional.one_hot(codes, 4).permute(0, 2, 1).float().contiguous()
        c = codes[:, DISTAL_RADIUS - LOCAL_RADIUS:DISTAL_RADIUS + LOCAL_RADIUS + 1]
        cat = (c[:, :-2] * 16 + c[:, 1:-1] * 4 + c[:, 2:]).contiguous()
        cont = torch.zeros(B, 1, device=device, dtype=torch.float64)
        dt = timed(lambda: model((cont, cat), x), 10, 2)
        out["random_windows_dense_tensors"] = {"bases_per_s": B / dt, "batch": B, "ms_per_call": dt * 1e3,
                                               "input_GB_per_s": B * (4 * (2 * DISTAL_RADIUS + 1) * 4 + cat.shape[1] * 8) / dt / 1e9,
                                               "note": "B x 2001 i.i.d. bases per site as cat_x int64 + distal_x fp32 one-hot"}
        # the reference's default call: 16 sites per forward (commands/predict.py:90), dense tensors, no sync between calls
        calls = [(cont[i * 16:(i + 1) * 16], cat[i * 16:(i + 1) * 16].contiguous(), x[i * 16:(i + 1) * 16].contiguous()) for i in range(256)]
        it = iter(range(10 ** 9))

        def one():
            co, ca, xx = calls[next(it) % 256]
            return model((co, ca), xx)

        dt = timed(one, 2000, 50)
        out["batch16_dense_calls"] = {"bases_per_s": 16 / dt, "us_per_call": dt * 1e6,
                                      "note": "one forward per 16 sites, 2000 calls back to back (encoding check read one call late)"}
        # model_predict_m over a loader that yields 16-row batches: small batches are fused into one launch
        n_rows = 16 * 1024
        ys = torch.zeros(n_rows, 1)
        loader = [(ys[i:i + 16], cont[i:i + 16].cpu(), cat[i:i + 16].cpu(), x[i:i + 16].cpu()) for i in range(0, n_rows, 16)]
        crit = nn.CrossEntropyLoss(reduction="sum")
        model_predict_m(model, loader, crit, device, N_CLASS)      # warm: staging pinned, the 8192-row workspace allocated
        runs = []
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            model_predict_m(model, loader, crit, device, N_CLASS)
            torch.cuda.synchronize()
            runs.append(time.perf_counter() - t0)
        dt = sorted(runs)[1]
        out["model_predict_m_batch16_loader"] = {"bases_per_s": n_rows / dt, "rows": n_rows, "seconds_of_three_runs": runs,
                                                 "note": "HOST tensors in 16-row batches (what the reference's loader yields): windows classified "
                                                         "into one symbol byte per column by host threads (mural_host_dense_to_symbols), 2 KB per "
                                                         "site over PCIe instead of 32 KB, 8192-row launches through mural_snv_forward_symbols"}
        dev_loader = [tuple(t.to(device) for t in b) for b in loader]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        model_predict_m(model, dev_loader, crit, device, N_CLASS)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out["model_predict_m_batch16_device_loader"] = {"bases_per_s": n_rows / dt, "rows": n_rows,
                                                        "note": "the same 16-row batches already on the device: fused into 8192-row launches"}
        # long windows (the reference advertises inputs of up to 64 kb, CHANGELOG:13): per-window predict at distal_radius 4000 with
        # S-config weights.  The large tower's pooled row (534 columns) exceeds a wave's LDS image: its first conv stage runs on
        # segments with halo columns (MuralSnvModel::longwin), the rest on the fused kernels as usual
        R4 = 4000
        m4 = build_model(device, R4)
        B4 = 2048
        codes4 = torch.randint(0, 4, (B4, 2 * R4 + 1), device=device, generator=g)
        x4 = torch.nn.functional.one_hot(codes4, 4).permute(0, 2, 1).float().contiguous()
        c4 = codes4[:, R4 - LOCAL_RADIUS:R4 + LOCAL_RADIUS + 1]
        cat4 = (c4[:, :-2] * 16 + c4[:, 1:-1] * 4 + c4[:, 2:]).contiguous()
        cont4 = torch.zeros(B4, 1, device=device, dtype=torch.float64)
        dt = timed(lambda: m4((cont4, cat4), x4), 5, 2)
        fl = snv_flop_per_site(R4)
        out["long_window_R4000"] = {"bases_per_s": B4 / dt, "batch": B4, "ms_per_call": dt * 1e3, "flop_per_site": fl,
                                    "tflops": fl * B4 / dt / 1e12, "frac_of_fp32_peak": fl * B4 / dt / 1e12 / PEAK_FP32_MFMA_TFLOPS,
                                    "fused_kernels": bool(m4._fused_ok()),
                                    "note": "distal_radius 4000 (window 8001), S-config weights_init weights, dense one-hot input of 2048 "
                                            "random windows per call (262 MB, classified into symbols inside the timed region); algorithmic "
                                            "FLOP by SURVEY 8d's rule; 512 windows per call: 0.26 of the roof (launches of 128-384 workgroups)"}
        # the same windows through the PACKED entry (forward_packed: no one-hot tensor, no dense -> symbol pass), and longer ones
        del x4
        glen = 400_000
        gcodes = torch.randint(0, 4, (glen,), generator=torch.Generator().manual_seed(7)).numpy().astype(np.uint8)
        gp, gm = pack2(gcodes)
        from mural_amd.data import PackedGenome
        g4 = PackedGenome(gp, gm, glen, device)

        def packed_variant(mdl, R, nwin):
            idx = torch.arange(nwin, device=device, dtype=torch.int64)
            pos, strand = (idx * 131) % (glen - 2 * R - 2) + R + 1, (idx & 1).to(torch.uint8)
            dtp = timed(lambda: mdl.forward_packed(g4, pos, strand, LOCAL_RADIUS, LOCAL_ORDER), 5, 2)
            flr = snv_flop_per_site(R)
            return {"bases_per_s": nwin / dtp, "batch": nwin, "ms_per_call": dtp * 1e3, "flop_per_site": flr, "tflops": flr * nwin / dtp / 1e12,
                    "frac_of_fp32_peak": flr * nwin / dtp / 1e12 / PEAK_FP32_MFMA_TFLOPS, "fused_kernels": bool(mdl._fused_ok()),
                    "note": "windows decoded from the packed genome inside the kernels (forward_packed)"}
        out["long_window_R4000_packed"] = packed_variant(m4, R4, 2048)
        out["long_window_R4000_packed_512"] = packed_variant(m4, R4, 512)
        del m4
        for R_long, nwin in ((8000, 1024), (16000, 512)):
            try:
                ml = build_model(device, R_long)
                out["long_window_R%d_packed" % R_long] = packed_variant(ml, R_long, nwin)
                del ml
            except Exception as e:      # noqa: BLE001  (a radius the fused kernels refuse is reported, not fatal)
                out["long_window_R%d_packed" % R_long] = {"error": f"{type(e).__name__}: {e}"[:300]}
    return out


def dense_reuse(device, model, genome, sites, steps):
    """The same workload through the cross-position reuse path: every site's result equals the
    per-window result within 1e-5 (tests/test_gpu_reuse.py), but overlapping windows share their first conv stage."""
    with torch.no_grad():
        pos, strand = sites[0]
        model.forward_packed_reuse(genome, pos, strand, local_radius=LOCAL_RADIUS, local_order=LOCAL_ORDER)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for s in range(steps):
            pos, strand = sites[s % len(sites)]
            model.forward_packed_reuse(genome, pos, strand, local_radius=LOCAL_RADIUS, local_order=LOCAL_ORDER)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
    n = sites[0][0].shape[0]
    return {"bases_per_s": n / dt, "ms_per_step": dt * 1e3, "sites_per_step": n,
            "note": "NOT the headline: dense same-chromosome site list, both strands; pooled first-stage rows are evaluated once per base "
                    "and strand, per site only the window-edge columns and the short stages (csrc/snv_reuse.hip)"}


_DUP_KEYS = {".layer.1.": ".bn1.", ".layer.2.": ".conv1.", ".layer.4.": ".bn2.", ".layer.5.": ".conv2."}


def shipped_snv_model(device, name="snv_pretrained_human_AT.npz"):
    """Network2 with the shipped Homo_sapiens/SNV/AT weights (they travel inside the parity fixture tests/golden/<name>, which stores
    every tensor once: the ResBlock's double-registered keys are expanded here); (model, local_radius, distal_radius)."""
    from mural_amd.model import model_choice
    fx = np.load(os.path.join(ROOT, "tests", "golden", name))
    r, order, R, h1, h2, ch, k, n_class = [int(v) for v in fx["hp"][:8]]
    cfg = dict(local_radius=r, local_order=order, local_hidden1_size=h1, local_hidden2_size=h2, distal_radius=R, emb_dropout=0.1,
               local_dropout=0.1, CNN_kernel_size=k, CNN_out_channels=ch, distal_fc_dropout=0.25)
    common = dict(emb_dims=[(4 ** order + 1, 2)] * (2 * r + 1 - (order - 1)), n_cont=0, n_class=n_class, distal_order=1, in_channels=4)
    model = model_choice(2, cfg, common, "snv")
    sd = {}
    for key in model.state_dict():
        src = key
        for a, b in _DUP_KEYS.items():
            src = src.replace(a, b)
        sd[key] = torch.from_numpy(np.asarray(fx["w::" + src]))
    model.load_state_dict(sd)
    return model.to(device).eval(), r, order, R


def synth_config5_files(workdir, device, n_chrom=3, chrom_len=7_000_000, seed=7):
    """Synthetic inputs of the file-to-file run: a FASTA of `n_chrom` i.i.d. uniform ACGT chromosomes (60-column lines) and the BED of
    an AT-model site list -- every A as a '+' site, every T as a '-' site (one focal base after strand complement, as the reference
    requires), labels 0..3.  Generated on the device (random draws, site selection, and the BED text through the library's own row
    formatter in its BED6 layout), because the host side of a 20 M-line text file is minutes of numpy.  Returns (fasta, bed, rows)."""
    from mural_amd import _lib
    from mural_amd.predict import _name_table, _tsv_struct
    lib = _lib.lib()
    fa, bed = os.path.join(workdir, "genome.fa"), os.path.join(workdir, "sites.bed")
    gen = torch.Generator(device=device).manual_seed(seed)
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=device)
    rows = 0
    piece = 1 << 21
    with open(fa, "wb") as f, open(bed, "wb") as b:
        for c in range(n_chrom):
            name = "chr%d" % (c + 1)
            codes = torch.randint(0, 4, (chrom_len,), device=device, generator=gen)
            seq = lut[codes]
            whole = chrom_len // 60 * 60
            lines = torch.cat([seq[:whole].view(-1, 60), torch.full((whole // 60, 1), 10, dtype=torch.uint8, device=device)], dim=1)
            f.write(b">" + name.encode() + b"\n")
            f.write(lines.cpu().numpy().tobytes())
            if whole < chrom_len:
                f.write(seq[whole:].cpu().numpy().tobytes() + b"\n")
            pos = torch.nonzero((codes == 0) | (codes == 3)).flatten()
            strand = (codes[pos] == 3).to(torch.uint8)
            u = torch.rand(pos.shape[0], device=device, generator=gen)
            label = ((u < 0.03).to(torch.float32) * (1 + (u * 1e4).to(torch.int64) % 3).to(torch.float32)).contiguous()
            end = pos + 1
            names = _name_table([name])
            t = _tsv_struct(names, 1, None, 0, 0, 0, 0, None, False, 0, 0, None, 0)
            t.layout = 1
            bound = int(lib.mural_tsv_row_bound(C.byref(t)))
            text = torch.empty(piece * bound, dtype=torch.uint8, device=device)
            count = torch.zeros(1, dtype=torch.int64, device=device)
            ws = torch.empty(int(lib.mural_tsv_format_workspace_bytes(piece)) + 256, dtype=torch.uint8, device=device)
            for r0 in range(0, pos.shape[0], piece):
                m = min(piece, pos.shape[0] - r0)
                t.start, t.end = pos[r0:].data_ptr(), end[r0:].data_ptr()
                t.strand, t.label, t.n = strand[r0:].data_ptr(), label[r0:].data_ptr(), m
                _lib.check(lib.mural_tsv_format_device(C.byref(t), text.data_ptr(), text.numel(), count.data_ptr(), ws.data_ptr(), ws.numel(),
                                                      _lib.current_stream_ptr(device)))
                b.write(text[:int(count.item())].cpu().numpy().tobytes())
            rows += int(pos.shape[0])
    return fa, bed, rows


def _reset_peak_rss():
    try:
        with open("/proc/self/clear_refs", "w") as fh:
            fh.write("5")
        return True
    except OSError:
        return False


def _rss_kb(field="VmHWM"):
    with open("/proc/self/status") as fh:
        for ln in fh:
            if ln.startswith(field + ":"):
                return int(ln.split()[1])
    return None


def rank_share(model, r, order, fa, bed, rows, work, device, t_one_rank, rank=3, world=8):
    """ONE rank's share of a `world`-rank file-to-file run, measured on this GPU (predict_bed_sharded(emulate=(rank, world)) with a
    part-file sink): its 1 / world of the index scan, the parse of its block of every chromosome, the FASTA pack, its block's compute
    and -- for a chromosome whose rows already are in the table's order (`aligned_shards`; the synthetic inputs are) -- the focal-base
    check of its own groups and its own rows formatted as its slice of the table; for any other chromosome a gathered shard of full size
    (the other ranks' site columns beside copies of this rank's probabilities stand in for the collective), the bed_reader reorder,
    the focal check and the full sort of the gathered shard, and its slice of the table.  Host seconds spent standing in for the other
    ranks are excluded (`emulation_seconds`).  projected_speedup = t(1 rank) / t(share)."""
    from mural_amd.predict import HipShardForward, TsvSink, predict_bed_sharded
    best = None
    for _ in range(2):
        split = {}
        reset = _reset_peak_rss()
        rss0 = _rss_kb("VmRSS")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fwd = HipShardForward(model, fa, r, order, device=device, reuse=True)
        sink = TsvSink(os.path.join(work, "share.tsv"), parts=(rank, world))
        n = predict_bed_sharded(fwd, bed, sink=sink, collect=False, timings=split, emulate=(rank, world))
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        assert n == rows
        t_share = wall - split["emulation"]
        if best is None or t_share < best["seconds"]:
            best = {"seconds": t_share, "wall_seconds": wall, "emulation_seconds": split["emulation"],
                    "split_seconds": {k: v for k, v in split.items() if isinstance(v, float)},
                    "aligned_shards": split.get("aligned_shards", 0),
                    "pack_thread_busy": fwd.seconds["pack"], "pack_wait": fwd.seconds["pack_wait"],
                    "rss_before_kb": rss0, "peak_rss_kb": _rss_kb("VmHWM") if reset else None,
                    "part_bytes": os.path.getsize(os.path.join(work, "share.tsv.part%04d" % rank))}
            # what placing this rank's slices into the ONE table file costs at the end (TsvSink._close_parts: every rank copies its own
            # slices through a shared mapping of the table, side by side): this part copied to its place in a file of the table's size
            part = os.path.join(work, "share.tsv.part%04d" % rank)
            dst = os.path.join(work, "assembled.tsv")
            with open(dst, "wb") as fh:
                fh.truncate(best["part_bytes"] * world)
            t0 = time.perf_counter()
            TsvSink._copy_slices(dst, part, [(0, best["part_bytes"] * rank, best["part_bytes"])])
            best["assemble_seconds_per_rank"] = time.perf_counter() - t0
            os.unlink(dst)
        os.unlink(os.path.join(work, "share.tsv.part%04d" % rank))
    best.update({"rank": rank, "world": world, "one_rank_seconds": t_one_rank, "projected_speedup_at_%d" % world: t_one_rank / best["seconds"],
                 "projected_speedup_with_one_table_file": t_one_rank / (best["seconds"] + best["assemble_seconds_per_rank"]),
                 "note": "predict_bed_sharded(emulate=(%d, %d)) + TsvSink(parts=(%d, %d)); best of two; peak_rss_kb = VmHWM of this "
                         "process over the run (reset through /proc/self/clear_refs; it includes the bench's resident genome and "
                         "models); projected_speedup = one-rank seconds / share seconds with the table left as per-rank part files; "
                         "..._with_one_table_file adds the copy of this rank's slices into the one table file (all ranks copy side by "
                         "side through a shared mapping)" % (rank, world, rank, world)})
    return best


def config5_chr1(device, chrom_len=248_000_000):
    """`python bench.py --config5-chr1`: the file-to-file run on ONE chromosome of human chr1's size (every A / T a site: ~124 M rows,
    3.3 GB of BED text), one rank's share of an 8-rank run beside the one-rank run, with the peak host RSS of each."""
    import shutil
    import tempfile
    from mural_amd.predict import HipShardForward, TsvSink, predict_bed_sharded
    model, r, order, R = shipped_snv_model(device)
    need = 14 << 30
    shm = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) and shutil.disk_usage("/dev/shm").free > need else None
    with tempfile.TemporaryDirectory(prefix="mural_c5_", dir=shm) as work:
        t0 = time.perf_counter()
        fa, bed, rows = synth_config5_files(work, device, 1, chrom_len)
        t_gen = time.perf_counter() - t0
        torch.cuda.empty_cache()
        out = os.path.join(work, "pred.tsv")
        one = None
        for _ in range(2):
            split = {}
            reset = _reset_peak_rss()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fwd = HipShardForward(model, fa, r, order, device=device, reuse=True)
            n = predict_bed_sharded(fwd, bed, sink=TsvSink(out), collect=False, timings=split)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            assert n == rows
            if one is None or dt < one["seconds"]:
                one = {"seconds": dt, "rows_per_s": rows / dt, "split_seconds": {k: v for k, v in split.items() if isinstance(v, float)},
                       "aligned_shards": split.get("aligned_shards", 0), "pack_thread_busy": fwd.seconds["pack"], "peak_rss_kb": _rss_kb("VmHWM") if reset else None,
                       "table_bytes": os.path.getsize(out)}
        os.unlink(out)
        share = rank_share(model, r, order, fa, bed, rows, work, device, one["seconds"])
        sizes = {"fasta_bytes": os.path.getsize(fa), "bed_bytes": os.path.getsize(bed)}
    return {"workload": "config5 file-to-file, one chromosome of %d bases, every A / T a site" % chrom_len, "rows": rows, **sizes,
            "one_rank": one, "rank_share": share, "input_generation_seconds_untimed": t_gen,
            "files_in": "/dev/shm" if shm else "the temp directory"}


def config5_e2e(device, n_chrom=3, chrom_len=14_000_000):
    """BASELINE.json configs[4] at N = 1, file to file: FASTA + BED -> sorted '%.4g' prediction table (run_predict.py:188-239) through
    mural_amd.predict.predict_bed_sharded with the shipped Homo_sapiens/SNV/AT weights: C++ BED reader / row order / FASTA packer,
    device-side column ordering, cross-position reuse kernels on the dense site list, focal-base check kernel, device sort + row
    formatter, writer thread.  Everything between the two file names is inside the timed region; `split` gives the host wall-clock of
    each stage of the driving thread and the busy times of the packer / writer threads that run beside it."""
    import tempfile
    from mural_amd.predict import HipShardForward, TsvSink, predict_bed_sharded
    model, r, order, R = shipped_snv_model(device)
    shm = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
    with tempfile.TemporaryDirectory(prefix="mural_c5_", dir=shm) as work:
        t0 = time.perf_counter()
        fa, bed, rows = synth_config5_files(work, device, n_chrom, chrom_len)
        t_gen = time.perf_counter() - t0
        out = os.path.join(work, "pred.tsv")
        sizes = {"fasta_bytes": os.path.getsize(fa), "bed_bytes": os.path.getsize(bed)}

        def run(reuse, path):
            split = {}
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fwd = HipShardForward(model, fa, r, order, device=device, reuse=reuse)
            split["fasta_scan"] = time.perf_counter() - t0
            sink = TsvSink(path)
            n = predict_bed_sharded(fwd, bed, sink=sink, collect=False, timings=split)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            assert n == rows
            split.update({"pack_thread_busy": fwd.seconds["pack"], "pack_wait": fwd.seconds["pack_wait"]})
            split.update({"sink_" + k: v for k, v in sink.seconds.items()})
            split.update({"writer_thread_" + k: v for k, v in sink.writer_seconds().items()})
            return dt, split, fwd.reuse_sites

        run(True, out)                                    # warm-up: kernels, allocator pools, page cache of the inputs
        dt, split, reused = run(True, out)
        table_bytes = os.path.getsize(out)
        with open(out, "rb") as fh:
            head = fh.read(400).split(b"\n")[:3]
        dt_pw, split_pw, _ = run(False, out)

        # the sink alone, fed pre-computed device shards as fast as it accepts them (sort by start, '%.4g' formatter, copy-out, writer
        # thread): the ceiling of the table writer.  `one_of_8`: one rank's share of an 8-rank run in the part-file mode (every rank
        # sorts the gathered shard and formats / writes ITS eighth of the sorted rows; rank 0 only strings the parts together), i.e.
        # what bounds config 5 at N = 8 is rows / that time
        def sink_only(parts):
            g = torch.Generator(device=device).manual_seed(5)
            per = rows // n_chrom
            shards = []
            for c in range(n_chrom):
                start = torch.randperm(chrom_len, device=device, generator=g)[:per].to(torch.int64)
                prob = torch.rand(per, 4, device=device, generator=g)
                prob = prob / prob.sum(1, keepdim=True)
                shards.append({"chrom": "chr%d" % (c + 1), "start": start, "end": start + 1,
                               "strand": (start & 1).to(torch.uint8), "label": torch.zeros(per, device=device), "prob": prob, "n_class": 4})
            best = None
            for _ in range(2):
                path = os.path.join(work, "sink_only.tsv")
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                sink = TsvSink(path, parts=parts) if parts else TsvSink(path)
                for sh in shards:
                    sink(sh)
                sink.close()
                torch.cuda.synchronize()
                dt_s = time.perf_counter() - t0
                best = dt_s if best is None or dt_s < best else best
            return per * n_chrom, best

        n_s, t_all = sink_only(False)
        _, t_8 = sink_only((3, 8))
        share = rank_share(model, r, order, fa, bed, rows, work, device, dt)
    return {"rows_per_s": rows / dt, "rank_share": share, "rows": rows, "seconds": dt, "chromosomes": n_chrom, "bases_per_chromosome": chrom_len,
            "sites_through_reuse_kernels": reused, "table_bytes": table_bytes, **sizes, "split_seconds": split,
            "per_window_kernels": {"rows_per_s": rows / dt_pw, "seconds": dt_pw, "split_seconds": split_pw},
            "sink_only_rows_per_s": n_s / t_all,
            "sink_only": {"rows": n_s, "single_writer_seconds": t_all, "single_writer_rows_per_s": n_s / t_all,
                          "one_of_8_seconds": t_8, "rows_per_s_ceiling_at_8_ranks": n_s / t_8,
                          "note": "TsvSink fed pre-computed device shards (random starts, 4 classes); one_of_8 = TsvSink(parts=(3, 8)): "
                                  "the share of one rank of an 8-rank part-file run (full sort of the gathered shard, an eighth of "
                                  "the rows formatted and written); best of two"},
            "weights": "Homo_sapiens/SNV/AT (shipped checkpoint, via tests/golden/snv_pretrained_human_AT.npz)",
            "input_generation_seconds_untimed": t_gen, "table_head": [h.decode() for h in head],
            "note": "FASTA + BED -> sorted '%.4g' TSV, one process, one GPU; second of two identical runs (inputs and output in "
                    + ("/dev/shm" if shm else "the temp directory") + "); every A ('+') and T ('-') of 3 random chromosomes is a site"}


_CONTRACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                  "dtype", "data", "config", "collective_check", "roofline", "cpu_baseline")


def _dig(obj, *path):
    for k in path:
        if not isinstance(obj, dict) or k not in obj:
            return None
        obj = obj[k]
    return obj


def summarize(line):
    """The secondary legs' headline numbers in one small object (the legs themselves are kilobytes each)."""
    r3 = lambda v: None if v is None else (round(v, 3) if abs(v) < 100 else round(v, 1))      # noqa: E731
    tr, ind, var, c5 = line.get("train"), line.get("indel"), line.get("variants"), line.get("config5_e2e")
    out = {
        "train_steps_per_s": {"symbol_windows": r3(_dig(tr, "steps_per_s")), "dense_input": r3(_dig(tr, "steps_per_s_dense_input")),
                              "synchronised": r3(_dig(tr, "steps_per_s_synchronised")), "frac_mfma": r3(_dig(tr, "roofline", "frac_mfma")),
                              "launches_per_step": _dig(tr, "roofline", "launches_per_step"),
                              "hbm_gb_per_step": r3((_dig(tr, "roofline", "hbm_bytes_per_step") or 0) / 1e9) or None},
        "indel_positions_per_s": {"packed": r3(_dig(ind, "positions_per_s")), "dense_input": r3(_dig(ind, "positions_per_s_dense_input")),
                                  "frac_mfma": r3(_dig(ind, "roofline", "frac")),
                                  "hbm_mb_per_position": r3((_dig(ind, "roofline", "hbm_bytes_per_position") or 0) / 1e6) or None,
                                  "train_ms_per_step": r3(_dig(ind, "train", "ms_per_step"))},
        "dense_reuse_bases_per_s": r3(_dig(line, "dense_reuse", "bases_per_s")),
        "config5_rows_per_s": r3(_dig(c5, "rows_per_s")),
        "config5_projected_speedup_at_8": r3(_dig(c5, "rank_share", "projected_speedup_at_8")),
        "cpu_baseline_bases_per_s": {"model_only": r3(_dig(line, "cpu_baseline", "value")),
                                     "with_encode": r3(_dig(line, "cpu_baseline", "bases_per_s_with_encode"))},
    }
    if isinstance(var, dict):
        frac = {}
        for name, v in var.items():
            if isinstance(v, dict):
                for key in ("frac_of_fp32_peak", "bases_per_s", "rows_per_s"):
                    if key in v and isinstance(v[key], (int, float)):
                        frac.setdefault(name, {})[key] = r3(v[key])
        out["variants"] = frac
    return out


def order_line(line):
    """The secondary legs first, the contract's keys + roofline + cpu_baseline + a compact `summary` LAST: whoever keeps only the tail of
    this one long line keeps the headline whole."""
    out = {k: v for k, v in line.items() if k not in _CONTRACT_KEYS}
    out["summary"] = summarize(line)
    for k in _CONTRACT_KEYS:
        if k in line:
            out[k] = line[k]
    return out


def main():
    try:      # (a side-stream AccumulateGrad warning of the training leg is ~1 KB of stderr per run)
        torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)
    except AttributeError:
        pass
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=500_000, help="sites per rank per step")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak (default): every rank evaluates --batch sites per step and each step ends with an all_gather; strong: the "
                         "FIXED 10M-position job of the config is split into contiguous blocks (mural_amd.predict.shard_bounds), every "
                         "rank walks its block in --steps slices and ONE all_gather ends the pass (SURVEY.md section 8e)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-train", action="store_true", help="skip the short train-steps/s and INDEL measurements (N=1 only)")
    ap.add_argument("--config5-chr1", action="store_true",
                    help="instead of the bench line: the file-to-file run on one chr1-sized chromosome (one rank, and one rank's share of 8)")
    args = ap.parse_args()
    if args.config5_chr1:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X (no HIP device visible); there is no CPU product path")
        torch.cuda.set_device(0)
        print(json.dumps(config5_chr1(torch.device("cuda", 0))), flush=True)
        return

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if args.gpus != 1 or world != 1:
            raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no HIP device visible); there is no CPU product path")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=device)

    from mural_amd import _lib
    from mural_amd.data import PackedGenome

    codes = synthetic_genome(GENOME_SITES + 2 * DISTAL_RADIUS)
    packed, mask = pack2(codes)
    genome = PackedGenome(packed, mask, len(codes), device)
    model = build_model(device)

    B = args.batch
    total_steps = args.warmup + args.steps
    strong = args.scaling == "strong"
    from mural_amd.predict import all_gather_rows, shard_bounds, verify_gathered_rows
    fwd = lambda p, st: model.forward_packed(genome, p, st, local_radius=LOCAL_RADIUS, local_order=LOCAL_ORDER)      # noqa: E731
    if strong:
        # the fixed job: the 10M consecutive sites of the config; this rank's block, walked in --steps slices of equal size
        all_idx = torch.arange(GENOME_SITES, device=device, dtype=torch.int64)
        all_pos, all_strand = all_idx + DISTAL_RADIUS, (all_idx & 1).to(torch.uint8)
        blo, bhi = shard_bounds(GENOME_SITES, rank, world)
        B = (bhi - blo + args.steps - 1) // args.steps

        def step_sites(s):
            k = (s - args.warmup) % args.steps if s >= args.warmup else s % args.steps
            a, b = shard_bounds(bhi - blo, k, args.steps)
            return all_pos[blo + a:blo + b], all_strand[blo + a:blo + b]
    else:
        # site list of this rank: step s covers sites [ (s*world + rank)*B , +B ) of the 10M-site list (wraps around)
        def step_sites(s):
            first = ((s * world + rank) * B) % GENOME_SITES
            idx = (first + torch.arange(B, device=device, dtype=torch.int64)) % GENOME_SITES
            return idx + DISTAL_RADIUS, (idx & 1).to(torch.uint8)

    sites = [step_sites(s) for s in range(total_steps)]
    # weak scaling: every step ends with all ranks' rows on every rank -- the step's all-gather runs beside the next step's compute
    # (mural_amd.predict.OverlappedGather: two buffers; tests/test_dist_gloo.py); everything is waited for inside the timed region
    from mural_amd.predict import OverlappedGather
    gather = OverlappedGather(B, N_CLASS, torch.float32, device) if world > 1 and not strong else None
    gathered = None
    parts = []

    def one_step(s):
        nonlocal gathered
        pos, strand = sites[s]
        out = fwd(pos, strand)
        if strong:
            if s >= args.warmup:
                parts.append(out)
        elif world > 1:
            gathered = gather.result(gather.submit(out))
        return out

    full = None
    with torch.no_grad():
        for s in range(args.warmup):
            out = one_step(s)
        if gather is not None:
            gather.finish()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        _lib.check(_lib.lib().mural_profile_begin())
        t0 = time.perf_counter()
        for s in range(args.warmup, total_steps):
            out = one_step(s)
        if strong:                                   # ONE collective for the whole pass: every rank ends with all 10M rows
            full = all_gather_rows(torch.cat(parts), GENOME_SITES)
        if gather is not None:
            gather.finish()                          # the last steps' collectives end inside the timed region
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
    k_ms, k_n = C.c_double(0.0), C.c_int64(0)
    _lib.check(_lib.lib().mural_profile_end(C.byref(k_ms), C.byref(k_n)))

    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # The collective, checked: rank 0 recomputes a random sample of the gathered rows -- drawn over every rank's block -- alone
    # and compares (per-site results do not depend on batch composition: the expected difference is exactly 0).
    rccl = None
    with torch.no_grad():
        if strong:
            diff, rows, owners = verify_gathered_rows(fwd, all_pos, all_strand, full, sample=8192, seed=rank)
            rccl = {"rccl_ranks": world, "rows_checked": rows, "blocks_touched": owners, "max_abs_diff": diff, "ok": diff == 0.0}
        elif world > 1:
            s_last = total_steps - 1
            firsts = [((s_last * world + r) * B) % GENOME_SITES for r in range(world)]
            idx = torch.cat([(f + torch.arange(B, device=device, dtype=torch.int64)) % GENOME_SITES for f in firsts])
            diff, rows, owners = verify_gathered_rows(fwd, idx + DISTAL_RADIUS, (idx & 1).to(torch.uint8), gathered, sample=8192, seed=rank)
            rccl = {"rccl_ranks": world, "rows_checked": rows, "blocks_touched": owners, "max_abs_diff": diff, "ok": diff == 0.0}
        if rccl is not None and world > 1:
            flag = torch.tensor([0.0 if rccl["ok"] else 1.0], device=device)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            rccl["ok_on_every_rank"] = bool(flag.item() == 0.0)
    if rccl is not None and not rccl["ok"]:
        print(f"bench.py: gathered rows differ from rank {rank}'s own evaluation by {rccl['max_abs_diff']:.3e}", file=sys.stderr, flush=True)

    # sanity: outputs are log-probabilities
    probs = out[:1024].exp().sum(dim=1)
    assert torch.allclose(probs, torch.ones_like(probs), atol=1e-4), "outputs are not normalised log-probabilities"

    if rank == 0:
        bases = GENOME_SITES if strong else args.steps * B * world
        kernel_ms = k_ms.value / max(k_n.value, 1)
        # the library launches the kernel once per (tower, stage-phase) pair -- the first stages per chunk of <= 131072 sites, the short
        # stages per four chunks -- each with its own tile size: sites_per_launch is the per-launch SHARE of the sites, so that FLOP_TOWERS x sites_per_launch / avg_launch_ms
        # = (all tower FLOP of the timed region) / (all tower-kernel time of the timed region)
        sites_per_launch = ((bhi - blo) if strong else args.steps * B) / max(k_n.value, 1)
        achieved = FLOP_TOWERS * sites_per_launch / (kernel_ms * 1e-3) / 1e12 if kernel_ms > 0 else 0.0
        # HBM bytes per launch: PMC counters need their own profiler passes, so the figure comes from the committed summary of
        # those passes over this same command (tools/profile_bench.sh -> profiles/hbm_traffic.json), scaled to this run's launches
        traffic, traffic_source = None, None
        fact = profile_fact("r06_predict") or profile_fact("r05_predict")
        if fact and fact.get("tower_hbm_bytes_per_site"):
            traffic = fact["tower_hbm_bytes_per_site"] * sites_per_launch
            traffic_source = "profiles/%s.json: %s" % ("r06_predict" if profile_fact("r06_predict") else "r05_predict", fact.get("source", ""))
        pmc = (fact or {}).get("all_launches", {})
        line = {
            "metric": "predicted bases/s (SNV local=10/distal=1000, 4-class, predict)",
            "value": bases / elapsed, "unit": "bases/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "SNV default predict, synthetic 10M positions (BASELINE.json configs[1])",
                       "local_radius": LOCAL_RADIUS, "distal_radius": DISTAL_RADIUS, "n_class": N_CLASS,
                       "sites_per_step_per_gpu": B, "input": "2-bit packed genome resident in HBM, every site's full window",
                       "weights": "weights_init, torch.manual_seed(0)", "parallelism": f"dp{world}",
                       "collective": ("one all_gather of the (10M, 4) result at the end of the pass" if strong else
                                      ("all_gather per step" if world > 1 else "none"))},
            "collective_check": rccl,
            "roofline": {"bound": "mfma", "kernel": _lib.lib().mural_snv_kernel_name().decode(),
                         "achieved": achieved, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_FP32_MFMA_TFLOPS, "traffic": traffic, "traffic_source": traffic_source,
                         "flop_per_launch": FLOP_TOWERS * sites_per_launch, "sites_per_launch": sites_per_launch,
                         "avg_launch_ms": kernel_ms, "launches": int(k_n.value),
                         # from the committed PMC passes (profiles/r06_predict.json), not measured in this run:
                         "profiled_mfma_pipe_busy": pmc.get("mfma_pipe_busy"), "profiled_held_clock_ghz": pmc.get("held_clock_ghz"),
                         "profiled_valu_per_mfma": pmc.get("valu_insts_per_mfma_excl_mfma"),
                         "note": "four launches per chunk of <= 131072 sites: (large | mid tower) x (first conv stage | the two short "
                                 "stages + fc + head); sites_per_launch / flop_per_launch / traffic are per-launch averages. "
                                 "Algorithmic FLOP of the layers this kernel evaluates (6,353,408 per site: every 32->32 "
                                 "conv + fc of both towers); the 1,691,136 FLOP/site of the two first conv layers are table "
                                 "lookups in snv_stage1_kernel and the 51,600 FLOP/site local MLP is snv_local_mlp; "
                                 "end-to-end model FLOP rate = 8,096,144 x value"},
        }
        def leg(name, fn):
            # the secondary objects must not cost the headline its line: a failure is reported in place of the object
            try:
                if os.environ.get("MURAL_BENCH_EMPTY_CACHE", "1") == "1":      # every leg starts from fresh device allocations
                    import gc
                    gc.collect()
                    torch.cuda.empty_cache()
                line[name] = fn()
            except Exception as e:      # noqa: BLE001
                line[name] = {"error": f"{type(e).__name__}: {e}"[:500]}
                print(f"bench.py: the '{name}' leg failed: {e!r}", file=sys.stderr, flush=True)

        if world == 1 and not args.no_train:
            leg("variants", lambda: workload_variants(device, model, genome))

            def reuse_leg():
                r = dense_reuse(device, model, genome, sites, max(2, min(args.steps, 10)))
                r["speedup_vs_per_window"] = r["bases_per_s"] / line["value"]
                return r
            # the training leg runs 3 % slower once the file-to-file leg (host threads, 20 GB of cached device blocks, pinned staging) ran
            # in the same process (603 vs 619 steps/s, same box, same build): the two step-rate legs go first.  MURAL_BENCH_LEG_ORDER
            # reorders them for A/B runs.
            order = os.environ.get("MURAL_BENCH_LEG_ORDER", "train,indel,reuse,config5").split(",")
            legs = {"reuse": lambda: leg("dense_reuse", reuse_leg), "config5": lambda: leg("config5_e2e", lambda: config5_e2e(device)),
                    "train": lambda: leg("train", lambda: train_steps_per_s(device, genome)),
                    "indel": lambda: leg("indel", lambda: indel_positions_per_s(device, genome))}
            for name in order:
                legs[name]()
        if world == 1 and not args.no_cpu_baseline:
            leg("cpu_baseline", lambda: cpu_baseline({k: v.detach().cpu() for k, v in model.state_dict().items()}, codes))
        print(json.dumps(order_line(line)), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
