"""Sharded genome-wide prediction: one process per GPU, sites split into contiguous blocks, ONE RCCL all_gather of the
per-rank log-probabilities per call (the reference itself is single-process and only advises to
split the BED file by hand, MuRaL/commands/predict.py:134-137).

The host logic (block partition, padded all_gather, trimming back to the reference's row order) is backend-agnostic
and covered by world_size-2 gloo tests on CPU; the compute function is the HIP model's ``forward_packed``.
"""
import numpy as np
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


# ------------------------------------------------------------------------------------------------------------------
# File-level sharded prediction (BASELINE config 5: whole-genome predict on 8 ranks).  Counterpart of the reference's advice
# to split a big BED by hand and run several `predict` processes (MuRaL/commands/predict.py:134-137) around the loop of
# MuRaL/scripts/run_predict.py:188-239.
#
#   * rows are processed in bed_reader order (preprocessing.py:39-106), one SHARD = one run of rows on the same chromosome;
#   * per shard only that chromosome is packed and resident in HBM (<= 70 MB for a human chromosome), and it is dropped
#     before the next one is loaded (per-chromosome streaming);
#   * rank i evaluates the contiguous block shard_bounds(rows of the shard, i, world): only its own block of sites is ever
#     uploaded to its device;
#   * ONE all_gather per shard returns (rows, n_class + 1) fp32: the probabilities and the strand-complemented focal base
#     that the reference's per-(segment, strand) consistency check needs (preprocessing.py:479-484, always run by
#     prepare_local_data :400 with local_order=1) -- groups may straddle rank boundaries, so the check runs on the gathered
#     shard, before the next shard starts;
#   * a sink on rank 0 (e.g. TsvSink) consumes each gathered shard; nothing of size N lives on a device.
# ------------------------------------------------------------------------------------------------------------------
def shard_runs(chrom_id):
    """[(lo, hi)] runs of equal chromosome id in a row sequence (the shards of predict_bed_sharded)."""
    chrom_id = np.asarray(chrom_id)
    if len(chrom_id) == 0:
        return []
    cut = np.r_[0, np.nonzero(chrom_id[1:] != chrom_id[:-1])[0] + 1, len(chrom_id)]
    return list(zip(cut[:-1].tolist(), cut[1:].tolist()))


def check_focal_groups(focal, group):
    """The reference's 'different bases' check: every (segment, strand) group of bed_reader shares one focal base after
    strand complement.  `group` ids are non-decreasing.  Raises ValueError (the reference exits)."""
    focal, group = np.asarray(focal), np.asarray(group)
    if len(focal) == 0:
        return
    first = np.r_[True, group[1:] != group[:-1]]
    ref = focal[np.maximum.accumulate(np.where(first, np.arange(len(focal)), 0))]
    if (focal != ref).any():
        raise ValueError("The positions in input BED file have different bases (A/T and C/G mixed)! The ref_genome or "
                         "input BED file could be wrong.")


class HipShardForward:
    """Default compute of predict_bed_sharded: packs the shard's chromosome from the FASTA file (C++ packer), keeps exactly one
    chromosome resident, runs the fused packed-genome forward in batches and returns softmax probabilities with the focal
    base appended as the last column."""

    def __init__(self, model, fasta_path, local_radius, local_order=3, distal_radius=None, device="cuda", batch_sites=1 << 20,
                 model_type="snv", dirichlet_weights=None, poisson=False, scale_factor=None):
        """`dirichlet_weights` / `poisson` / `scale_factor`: apply the post-head calibration chain of run_predict.py:217-225
        (and scripts/scaling.py) on the device, fused behind the head (calibration.calibrate_device); the shard then carries
        float64 calibrated probabilities and the sink must not calibrate again."""
        from .data import ingest
        self.calibration = dict(dirichlet_weights=dirichlet_weights, poisson=poisson, scale_factor=scale_factor)
        self.calibrated = dirichlet_weights is not None or bool(poisson) or bool(scale_factor)
        self._ingest = ingest
        self.model = model.to(device).eval()
        self.fasta_path, self.device = fasta_path, torch.device(device)
        self.local_radius, self.local_order, self.distal_radius = local_radius, local_order, distal_radius
        self.batch_sites, self.model_type = batch_sites, model_type
        self.records = {r.name: r for r in ingest.scan_fasta(fasta_path)}
        self._resident = (None, None)

    def genome(self, chrom):
        if self._resident[0] != chrom:
            self._resident = (None, None)              # drop the previous chromosome before the next one is uploaded
            if chrom not in self.records:
                raise KeyError(chrom)                  # the reference's seq_records[chrom] lookup
            packed, mask, n, amb = self._ingest.pack_fasta_record(self.fasta_path, self.records[chrom])
            from .data.genome import PackedGenome
            self._resident = (chrom, PackedGenome(packed, mask, n, self.device, amb))
        return self._resident[1]

    @torch.no_grad()
    def __call__(self, chrom, pos, strand):
        g = self.genome(chrom)
        n = len(pos)
        out = torch.empty((n, self.model.n_class + 1), dtype=torch.float64 if self.calibrated else torch.float32,
                          device=self.device)
        for r0 in range(0, n, self.batch_sites):
            p = torch.from_numpy(pos[r0:r0 + self.batch_sites]).to(self.device)
            st = torch.from_numpy(strand[r0:r0 + self.batch_sites]).to(self.device)
            if self.model_type == "snv":
                logp = self.model.forward_packed(g, p, st, local_radius=self.local_radius, local_order=self.local_order)
                out[r0:r0 + len(p), -1] = g.encode_kmer(p, st, 1, 1)[:, 1].to(torch.float32)   # complemented focal base 0..4
            else:
                logp = self.model.forward_packed(g, p, st, self.distal_radius)
                out[r0:r0 + len(p), -1] = 0.0
            if self.calibrated:
                from .calibration import calibrate_device
                out[r0:r0 + len(p), :-1] = calibrate_device(logp, **self.calibration)
            else:
                out[r0:r0 + len(p), :-1] = torch.softmax(logp, dim=1)
        return out


class TsvSink:
    """Rank-0 consumer of gathered shards: the prediction table of run_predict.py:217-239 (optional Dirichlet / Poisson
    calibration, columns chrom start end strand mut_type prob0.., rows sorted by (chrom, start), '%.4g').  Shards whose
    chromosomes arrive in ascending name order are sorted by start and appended immediately (nothing of size N is kept);
    otherwise the rows are buffered and sorted at close()."""

    def __init__(self, path, poisson=False, dirichlet_weights=None):
        self.path, self.poisson, self.dirichlet_weights = path, poisson, dirichlet_weights
        self._buffer, self._streaming, self._last, self._wrote_header = [], True, None, False
        open(path, "w").close()

    def _frame(self, shard):
        import pandas as pd
        prob = np.asarray(shard["prob"])
        if self.dirichlet_weights is not None:
            from .calibration import dirichlet_calibrate
            prob = dirichlet_calibrate(prob, self.dirichlet_weights)
        if self.poisson:
            from .data.ingest import poisson_calibrate
            prob = poisson_calibrate(prob)
        cols = {"chrom": shard["chrom"], "start": shard["start"], "end": shard["end"], "strand": shard["strand"],
                "mut_type": np.asarray(shard["label"]).astype(np.int64)}
        cols.update({"prob%d" % i: prob[:, i] for i in range(prob.shape[1])})
        return pd.DataFrame(cols)

    def _write(self, df):
        df.to_csv(self.path, sep="\t", float_format="%.4g", index=False, mode="a", header=not self._wrote_header)
        self._wrote_header = True

    def __call__(self, shard):
        name = str(shard["chrom"][0]) if len(shard["chrom"]) else None
        if self._streaming and name is not None and (self._last is None or name > self._last):
            self._last = name
            df = self._frame(shard)
            df.sort_values(["start"], inplace=True, kind="stable")
            self._write(df)
        else:
            self._streaming = False
            self._buffer.append(self._frame(shard))

    def close(self):
        import pandas as pd
        if self._buffer:
            if self._wrote_header:                   # some shards were already streamed out: merge them back in
                self._buffer.insert(0, pd.read_csv(self.path, sep="\t"))
                open(self.path, "w").close()
                self._wrote_header = False
            df = pd.concat(self._buffer, ignore_index=True)
            df.sort_values(["chrom", "start"], inplace=True, kind="stable")
            self._write(df)
            self._buffer = []
        elif not self._wrote_header:
            self._write(self._frame({"chrom": np.zeros(0, object), "start": np.zeros(0, np.int64), "end": np.zeros(0, np.int64),
                                     "strand": np.zeros(0, object), "label": np.zeros(0), "prob": np.zeros((0, 0))}))


def predict_bed_sharded(forward, bed_path, segment_center=300000, model_type="snv", group=None, sink=None, collect=True):
    """Sharded file-level prediction.  `forward(chrom_name, pos, strand) -> (rows, n_class + 1)` tensor (probabilities + focal
    base; see HipShardForward) is called once per shard with THIS rank's block of the shard's sites.  Every rank takes part in
    one all_gather per shard; `sink(shard_dict)` is called on rank 0 with the gathered shard (keys chrom, start, end, strand,
    label, prob, order).  With `collect` the function also returns those arrays for ALL rows (bed_reader order) on every rank --
    leave it off for genome-scale inputs and let the sink stream them out.  Returns the dict (or row count if not collecting)."""
    from .data import ingest
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    sites = ingest.read_bed(bed_path)
    order, grp = ingest.bed_order(sites, segment_center)
    cid, start, strand = sites.chrom_id[order], sites.start[order], sites.strand[order]
    names = np.asarray(sites.chrom_names, dtype=object)
    kept = []
    for lo, hi in shard_runs(cid):
        chrom = sites.chrom_names[cid[lo]]
        b0, b1 = shard_bounds(hi - lo, rank, world)
        local = forward(chrom, start[lo + b0:lo + b1], strand[lo + b0:lo + b1])
        if local.shape[0] != b1 - b0:
            raise RuntimeError("forward returned a wrong number of rows")
        full = all_gather_rows(local, hi - lo, group).cpu().numpy()
        if model_type == "snv":
            check_focal_groups(full[:, -1].astype(np.int64), grp[lo:hi])
        shard = {"chrom": names[cid[lo:hi]], "start": start[lo:hi], "end": sites.end[order[lo:hi]],
                 "strand": np.where(strand[lo:hi] == 1, "-", "+"), "label": sites.score[order[lo:hi]], "prob": full[:, :-1],
                 "order": order[lo:hi]}
        if sink is not None and rank == 0:
            sink(shard)
        if collect:
            kept.append(shard)
    if sink is not None and rank == 0 and hasattr(sink, "close"):
        sink.close()
    if not collect:
        return len(order)
    if not kept:
        return {"chrom": np.zeros(0, object), "start": np.zeros(0, np.int64), "end": np.zeros(0, np.int64),
                "strand": np.zeros(0, object), "label": np.zeros(0, np.float32), "prob": np.zeros((0, 0), np.float32),
                "order": np.zeros(0, np.int64)}
    return {k: np.concatenate([sh[k] for sh in kept]) for k in kept[0]}
