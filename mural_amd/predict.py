"""Sharded genome-wide prediction: one process per GPU, sites split into contiguous blocks, ONE RCCL all_gather of the
per-rank log-probabilities per call (the reference itself is single-process and only advises to
split the BED file by hand, MuRaL/commands/predict.py:134-137).

The host logic (block partition, padded all_gather, trimming back to the reference's row order) is backend-agnostic
and covered by world_size-2 gloo tests on CPU; the compute function is the HIP model's ``forward_packed``.
"""
import torch
import torch.distributed as dist


def shard_bounds(n, rank, world):
    """Contiguous block [lo, hi) of `n` rows for `rank`: the first n % world ranks take one extra row."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world {rank}/{world}")
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def all_gather_rows(local, n_total, group=None):
    """Gather row blocks of unequal length (block partition of shard_bounds) into the full (n_total, C) tensor on
    every rank with a single all_gather_into_tensor of equally padded blocks."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        if local.shape[0] != n_total:
            raise ValueError("single-process gather expects all rows")
        return local
    width = local.shape[1]
    per = (n_total + world - 1) // world                       # longest block
    padded = torch.zeros((per, width), dtype=local.dtype, device=local.device)
    padded[: local.shape[0]] = local
    gathered = torch.empty((world * per, width), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(gathered, padded, group=group)
    out = torch.empty((n_total, width), dtype=local.dtype, device=local.device)
    for r in range(world):
        lo, hi = shard_bounds(n_total, r, world)
        out[lo:hi] = gathered[r * per: r * per + (hi - lo)]
    return out


def predict_sites(forward_fn, pos, strand, group=None):
    """Run `forward_fn(pos_block, strand_block) -> (rows, n_class)` on this rank's block of sites and return the
    full (N, n_class) result in input order on every rank.  `pos` / `strand` hold ALL sites on every rank (they are
    8 + 1 bytes per site; the genome and the weights are replicated)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    n = pos.shape[0]
    lo, hi = shard_bounds(n, rank, world)
    local = forward_fn(pos[lo:hi], strand[lo:hi])
    if local.shape[0] != hi - lo:
        raise RuntimeError("forward_fn returned a wrong number of rows")
    return all_gather_rows(local, n, group)


class ShardedPredictor:
    """Convenience wrapper binding a HIP model and a resident PackedGenome."""

    def __init__(self, model, genome, local_radius, local_order=3, group=None):
        self.model, self.genome = model, genome
        self.local_radius, self.local_order, self.group = local_radius, local_order, group

    @torch.no_grad()
    def __call__(self, pos, strand):
        fn = lambda p, s: self.model.forward_packed(self.genome, p, s, self.local_radius, self.local_order)
        return predict_sites(fn, pos, strand, self.group)
