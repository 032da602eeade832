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

Extra objects on the JSON line: ``train`` (N=1: steps/s of the S-config training step at batch 4096, BASELINE.json
configs[2], measured after the timed prediction region); ``indel`` (N=1: UNet_Small positions/s, configs[3]); ``roofline`` (dominant kernel = snv_towers_fused: algorithmic FLOP per launch of the
layers it evaluates / HIP-event duration of that kernel measured live in the timed region, vs the 157.3 TFLOP/s fp32
MFMA peak) and
``cpu_baseline`` (oracle = PyTorch-CPU restatement of the reference, timed on this box's host cores on a bounded
sample of the same workload; rank 0, N=1 only).
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


def build_model(device):
    from mural_amd.model import model_choice, weights_init
    ncol = 2 * LOCAL_RADIUS + 1 - (LOCAL_ORDER - 1)
    cfg = dict(local_radius=LOCAL_RADIUS, local_order=LOCAL_ORDER, local_hidden1_size=150, local_hidden2_size=75,
               distal_radius=DISTAL_RADIUS, emb_dropout=0.1, local_dropout=0.1, CNN_kernel_size=3, CNN_out_channels=32,
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


def cpu_baseline(model_state, codes, budget_s=12.0, batch=256):
    """Time the oracle (CPU restatement of the reference, oracle/snv_ref.py) on a bounded sample of the workload."""
    from oracle import encode_ref, snv_ref
    orc = snv_ref.build(2, local_radius=LOCAL_RADIUS, local_order=LOCAL_ORDER, distal_radius=DISTAL_RADIUS)
    orc.load_state_dict(model_state)
    orc.eval()
    cont = torch.zeros(batch, 1, dtype=torch.float64)

    def batch_inputs(it):
        pos = DISTAL_RADIUS + it * batch + np.arange(batch)
        sym = ["-" if (p - DISTAL_RADIUS) % 2 else "+" for p in pos]
        return (torch.from_numpy(encode_ref.kmer_encode(codes, pos, sym, LOCAL_RADIUS, LOCAL_ORDER)),
                torch.from_numpy(encode_ref.onehot_encode(codes, pos, sym, DISTAL_RADIUS)))

    def timed(it):
        cat, x = batch_inputs(it)
        t0 = time.perf_counter()
        orc((cont, cat), x)
        return time.perf_counter() - t0

    with torch.no_grad():
        # intra-op threading of the small convs does not scale to every core of a big host: pick the best of a few
        # thread counts (2 untimed + 2 timed iterations each), then time that setting for the budget
        ncpu = os.cpu_count() or 1
        best_threads, best_dt = 1, float("inf")
        for threads in sorted({min(ncpu, t) for t in (8, 16, 32, 64)}):
            torch.set_num_threads(threads)
            timed(0), timed(1)
            dt = min(timed(2), timed(3))
            if dt < best_dt:
                best_threads, best_dt = threads, dt
        torch.set_num_threads(best_threads)
        timed(0)
        done, t_used, it = 0, 0.0, 0
        while t_used < budget_s and it < 400:
            t_used += timed(4 + it)
            done += batch
            it += 1
    # the training step of the same restatement beside it: batch 128 (the reference's default), CE(sum) + clip + Adam
    orc.train()
    tb = 128
    opt = torch.optim.Adam(orc.parameters(), lr=1e-3)
    crit = torch.nn.CrossEntropyLoss(reduction="sum")
    cat, x = batch_inputs(0)
    cat, x, y = cat[:tb], x[:tb], torch.from_numpy(np.arange(tb) % N_CLASS)

    def train_step():
        t0 = time.perf_counter()
        loss = crit(orc((cont[:tb], cat), x), y)
        opt.zero_grad()
        loss.backward()
        torch.nn.utils.clip_grad_norm_(orc.parameters(), max_norm=10)
        opt.step()
        return time.perf_counter() - t0

    train_step()
    t_train = min(train_step(), train_step())
    return {"value": done / max(t_used, 1e-9), "unit": "bases/s", "cores": best_threads, "kind": "port",
            "sample": f"{done} sites of the same workload, model only (inputs pre-encoded), batch {batch}, "
                      f"{it} timed iterations after warm-up, best of 8/16/32/64 torch threads on {ncpu} host CPUs",
            "train_steps_per_s": 1.0 / t_train, "train_batch": tb, "train_sites_per_s": tb / t_train}


def train_steps_per_s(device, genome, B=4096, steps=30, warmup=5):
    """BASELINE.json configs[2]: S-config from scratch, batch 4096, Adam lr 1e-3, CE-sum, clip 10, default dropouts."""
    import torch.nn as nn
    model = build_model(device).train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    crit = nn.CrossEntropyLoss(reduction="sum")
    rng = np.random.default_rng(1)
    labels = torch.from_numpy(rng.choice(4, size=(steps + warmup) * B, p=[0.955, 0.015, 0.015, 0.015])).to(device)
    cont = torch.zeros(B, 1, device=device)
    times = []
    for s in range(steps + warmup):
        idx = torch.arange(s * B, (s + 1) * B, device=device)
        pos, strand = idx + DISTAL_RADIUS, (idx & 1).to(torch.uint8)
        cat = genome.encode_kmer(pos, strand, LOCAL_RADIUS, LOCAL_ORDER)
        x = genome.encode_onehot(pos, strand, DISTAL_RADIUS)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        loss = crit(model((cont, cat), x), labels[s * B:(s + 1) * B])
        opt.zero_grad()
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 10)
        opt.step()
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    t = float(np.median(times[warmup:]))
    return {"steps_per_s": 1.0 / t, "ms_per_step": t * 1e3, "batch": B, "sites_per_s": B / t, "optimizer": "Adam lr 1e-3",
            "loss": "CrossEntropy(sum), clip_grad_norm 10",
            "note": "forward (batch-stat BN, dropout) + backward + update; inputs already encoded on the device; median of %d "
                    "individually synchronised steps after %d warm-up (the reference loop reads loss.item() every step; "
                    "tools/bench_variants.py train measures the un-synchronised loop)" % (steps, warmup)}


def indel_positions_per_s(device, genome, n=4096):
    """BASELINE.json configs[3]: UNet_Small, human-insertion geometry (L=8000, 8 classes, use_reverse), packed input."""
    from mural_amd.model import model_choice, weights_init
    cfg = dict(CNN_out_channels=8, CNN_kernel_size=7, down_list=[1, 4, 5, 5, 5, 2], use_reverse=True)
    torch.manual_seed(0)
    model = model_choice(0, cfg, dict(n_class=8), "indel")
    model.apply(weights_init)
    model = model.to(device).eval()
    idx = torch.arange(n, device=device, dtype=torch.int64)
    pos, strand = idx * 100 + 4000, (idx & 1).to(torch.uint8)
    with torch.no_grad():
        model.forward_packed(genome, pos, strand, 4000)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            model.forward_packed(genome, pos, strand, 4000)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
    out = {"positions_per_s": n / dt, "positions": n, "window": 8000, "n_class": 8, "algorithmic_TFLOPs": n / dt * 113.4e6 / 1e12,
           "note": "weights_init, window decode from the packed genome inside the timed region; 113.4 MFLOP/position"}
    # one training configuration of the same model: batch 128 (the reference's default), CE(sum) + clip + Adam
    tb = 128
    model.train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    crit = torch.nn.CrossEntropyLoss(reduction="sum")
    x = genome.encode_onehot(pos[:tb], strand[:tb], 4000, "indel")
    y = (idx[:tb] % 8)

    def step():
        loss = crit(model(x), y)
        opt.zero_grad()
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=10, error_if_nonfinite=False)
        opt.step()

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    out["train"] = {"steps_per_s": 1.0 / dt, "ms_per_step": dt * 1e3, "batch": tb, "positions_per_s": tb / dt,
                    "note": "forward (batch-statistics BatchNorm, dropout) + backward + clip + Adam on pre-encoded windows"}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=500_000, help="sites per rank per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-train", action="store_true", help="skip the short train-steps/s and INDEL measurements (N=1 only)")
    args = ap.parse_args()

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
    # site list of this rank: step s covers sites [ (s*world + rank)*B , +B ) of the 10M-site list (wraps around)
    def step_sites(s):
        first = ((s * world + rank) * B) % GENOME_SITES
        idx = (first + torch.arange(B, device=device, dtype=torch.int64)) % GENOME_SITES
        return idx + DISTAL_RADIUS, (idx & 1).to(torch.uint8)

    sites = [step_sites(s) for s in range(total_steps)]
    gathered = torch.empty((world * B, N_CLASS), dtype=torch.float32, device=device) if world > 1 else None

    def one_step(s):
        pos, strand = sites[s]
        out = model.forward_packed(genome, pos, strand, local_radius=LOCAL_RADIUS, local_order=LOCAL_ORDER)
        if world > 1:
            dist.all_gather_into_tensor(gathered, out)
        return out

    with torch.no_grad():
        for s in range(args.warmup):
            out = one_step(s)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        _lib.check(_lib.lib().mural_profile_begin())
        t0 = time.perf_counter()
        for s in range(args.warmup, total_steps):
            out = one_step(s)
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

    # sanity: outputs are log-probabilities
    probs = out[:1024].exp().sum(dim=1)
    assert torch.allclose(probs, torch.ones_like(probs), atol=1e-4), "outputs are not normalised log-probabilities"

    if rank == 0:
        bases = args.steps * B * world
        kernel_ms = k_ms.value / max(k_n.value, 1)
        # the library launches the kernel four times per chunk of <= 131072 sites ((tower, stage-phase) pairs, each with its own tile
        # size): sites_per_launch is the per-launch SHARE of the sites, so that FLOP_TOWERS x sites_per_launch / avg_launch_ms
        # = (all tower FLOP of the timed region) / (all tower-kernel time of the timed region)
        sites_per_launch = args.steps * B / max(k_n.value, 1)
        achieved = FLOP_TOWERS * sites_per_launch / (kernel_ms * 1e-3) / 1e12 if kernel_ms > 0 else 0.0
        traffic = None     # HBM bytes per launch from the committed PMC passes (tools/profile_bench.sh), if present
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tpath):
            with open(tpath) as fh:
                traffic = json.load(fh)["hbm_bytes_per_site"] * sites_per_launch
        line = {
            "metric": "predicted bases/s (SNV local=10/distal=1000, 4-class, predict)",
            "value": bases / elapsed, "unit": "bases/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "SNV default predict, synthetic 10M positions (BASELINE.json configs[1])",
                       "local_radius": LOCAL_RADIUS, "distal_radius": DISTAL_RADIUS, "n_class": N_CLASS,
                       "sites_per_step_per_gpu": B, "input": "2-bit packed genome resident in HBM, every site's full window",
                       "weights": "weights_init, torch.manual_seed(0)", "parallelism": f"dp{world}",
                       "collective": "all_gather per step" if world > 1 else "none"},
            "roofline": {"bound": "mfma", "kernel": _lib.lib().mural_snv_kernel_name().decode(),
                         "achieved": achieved, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_FP32_MFMA_TFLOPS, "traffic": traffic,
                         "flop_per_launch": FLOP_TOWERS * sites_per_launch, "sites_per_launch": sites_per_launch,
                         "avg_launch_ms": kernel_ms, "launches": int(k_n.value),
                         "note": "four launches per chunk of <= 131072 sites: (large | mid tower) x (first conv stage | the two short "
                                 "stages + fc + head); sites_per_launch / flop_per_launch / traffic are per-launch averages. "
                                 "Algorithmic FLOP of the layers this kernel evaluates (6,353,408 per site: every 32->32 "
                                 "conv + fc of both towers); the 1,691,136 FLOP/site of the two first conv layers are table "
                                 "lookups in snv_stage1_kernel and the 51,600 FLOP/site local MLP is snv_local_mlp; "
                                 "end-to-end model FLOP rate = 8,096,144 x value"},
        }
        if world == 1 and not args.no_train:
            line["train"] = train_steps_per_s(device, genome)
            line["indel"] = indel_positions_per_s(device, genome)
        if world == 1 and not args.no_cpu_baseline:
            state = {k: v.detach().cpu() for k, v in model.state_dict().items()}
            line["cpu_baseline"] = cpu_baseline(state, codes)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
