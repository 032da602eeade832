"""Sharded genome-wide prediction: one process per GPU, sites split into contiguous blocks, ONE RCCL all_gather of the
per-rank probabilities per call (the reference itself is single-process and only advises to
split the BED file by hand, MuRaL/commands/predict.py:134-137).

The host logic (block partition, padded all_gather, trimming back to the reference's row order) is backend-agnostic
and covered by world_size-2 gloo tests on CPU; the compute function is the HIP model's ``forward_packed`` /
``forward_packed_reuse``; the prediction table is formatted by ``csrc/tsv.hip`` (device kernel or host threads).
"""
import ctypes as C
import os
import queue
import threading
import time

import numpy as np
import torch
import torch.distributed as dist

from . import _lib


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
    if n_total == world * per:
        return gathered
    out = torch.empty((n_total, width), dtype=local.dtype, device=local.device)
    for r in range(world):
        lo, hi = shard_bounds(n_total, r, world)
        out[lo:hi] = gathered[r * per: r * per + (hi - lo)]
    return out


class OverlappedGather:
    """The per-step collective of a weak-scaling run (every rank contributes `rows` rows per step, every rank ends the step with all
    of them) issued asynchronously into one of TWO buffers: the all-gather of step s runs on the collective's stream while step s + 1
    is computed; a buffer is waited for when it comes up again and at ``finish()``.  Same bytes as a blocking
    ``all_gather_into_tensor`` per step; on xGMI the 64 MB of an 8-rank step then hide behind the next step's 29 ms of compute instead
    of adding to them."""

    def __init__(self, rows, width, dtype, device, group=None):
        self.group = group
        self.world = dist.get_world_size(group)
        self.bufs = [torch.empty((self.world * rows, width), dtype=dtype, device=device) for _ in range(2)]
        self.works = [None, None]
        self.turn = 0

    def submit(self, local):
        """Start gathering `local` (rows, width); returns the index of the buffer that will hold the step's rows."""
        i = self.turn
        self.turn ^= 1
        if self.works[i] is not None:
            self.works[i][0].wait()
        # (the source stays referenced until its collective was waited for)
        self.works[i] = (dist.all_gather_into_tensor(self.bufs[i], local.contiguous(), group=self.group, async_op=True), local)
        return i

    def finish(self):
        for w in self.works:
            if w is not None:
                w[0].wait()
        self.works = [None, None]

    def result(self, i):
        return self.bufs[i]


def predict_sites(forward_fn, pos, strand, group=None, steps=1):
    """Run `forward_fn(pos_block, strand_block) -> (rows, n_class)` on this rank's block of sites and return the
    full (N, n_class) result in input order on every rank.  `pos` / `strand` hold ALL sites on every rank (they are
    8 + 1 bytes per site; the genome and the weights are replicated).  The block is evaluated in `steps` slices (bounded
    workspace; the benchmark's timed steps) and ends with ONE all_gather of the whole block."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    n = pos.shape[0]
    lo, hi = shard_bounds(n, rank, world)
    parts = []
    for k in range(max(int(steps), 1)):
        a, b = shard_bounds(hi - lo, k, max(int(steps), 1))
        parts.append(forward_fn(pos[lo + a:lo + b], strand[lo + a:lo + b]))
    local = parts[0] if len(parts) == 1 else torch.cat(parts)
    if local.shape[0] != hi - lo:
        raise RuntimeError("forward_fn returned a wrong number of rows")
    return all_gather_rows(local, n, group)


def verify_gathered_rows(forward_fn, pos, strand, full, sample=4096, seed=0):
    """Recompute a random sample of rows of a gathered result on THIS rank alone and compare: (largest absolute difference,
    rows checked, ranks whose blocks the sample touched).  The sample is drawn over all rows, i.e. over every rank's block, so a
    collective that scrambles, drops or pads blocks shows up here; per-site results do not depend on batch composition, so the
    expected difference is exactly 0."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    n = pos.shape[0]
    if n == 0:
        return 0.0, 0, 0
    g = torch.Generator().manual_seed(seed)
    idx = torch.randperm(n, generator=g)[:min(sample, n)].sort().values
    dev_idx = idx.to(pos.device)
    mine = forward_fn(pos[dev_idx], strand[dev_idx])
    diff = float((mine.double() - full[dev_idx].double()).abs().max())
    owners = {r for r in range(world) if ((idx >= shard_bounds(n, r, world)[0]) & (idx < shard_bounds(n, r, world)[1])).any()}
    return diff, int(idx.numel()), len(owners)


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
#   * rows keep their bed_reader order (preprocessing.py:39-106); one SHARD = all rows of one chromosome (its runs in that
#     order, concatenated), and the shards are processed in ascending chromosome NAME order -- the order of the final
#     sort_values(['chrom', 'start']) -- so the table can be streamed out shard by shard;
#   * per shard only that chromosome is packed and resident in HBM (<= 70 MB for a human chromosome); the next chromosome is
#     packed on a host thread while this one is computed, and every chromosome is packed exactly once;
#   * rank i evaluates the contiguous block shard_bounds(rows of the shard, i, world), through the cross-position reuse kernels
#     where the block's sites are dense along the chromosome and through the per-window kernels elsewhere;
#   * ONE all_gather per shard returns (rows, n_class + 1): the probabilities and the strand-complemented focal base that the
#     reference's per-(segment, strand) consistency check needs (preprocessing.py:479-484, always run by prepare_local_data
#     :400 with local_order=1) -- groups may straddle rank boundaries, so the check (a streaming kernel) runs on the gathered
#     shard; its verdict is read one shard late, so no rank waits for it;
#   * rank 0 hands the gathered shard to a sink; TsvSink sorts it by start on the device, formats the text rows on the device
#     (csrc/tsv.hip) and leaves the copy to the host and the write() to a writer thread: no rank waits for the writer unless
#     all of its text buffers are full.
# With a host-memory forward (gloo ranks, CPU tests) the same driver runs on numpy arrays and the host formatter.
# ------------------------------------------------------------------------------------------------------------------
def shard_runs(chrom_id):
    """[(lo, hi)] runs of equal chromosome id in a row sequence."""
    chrom_id = np.asarray(chrom_id)
    if len(chrom_id) == 0:
        return []
    cut = np.r_[0, np.nonzero(chrom_id[1:] != chrom_id[:-1])[0] + 1, len(chrom_id)]
    return list(zip(cut[:-1].tolist(), cut[1:].tolist()))


_FOCAL_MSG = ("The positions in input BED file have different bases (A/T and C/G mixed)! The ref_genome or "
              "input BED file could be wrong.")


def check_focal_groups(focal, group):
    """The reference's 'different bases' check: every (segment, strand) group of bed_reader shares one focal base after
    strand complement.  `group` ids are non-decreasing.  Raises ValueError (the reference exits)."""
    focal, group = np.asarray(focal), np.asarray(group)
    if len(focal) == 0:
        return
    first = np.r_[True, group[1:] != group[:-1]]
    ref = focal[np.maximum.accumulate(np.where(first, np.arange(len(focal)), 0))]
    if (focal != ref).any():
        raise ValueError(_FOCAL_MSG)


class HipShardForward:
    """Default compute of predict_bed_sharded: packs the shard's chromosome from the FASTA file (C++ packer; the next one on a
    host thread while this one is computed), keeps exactly one chromosome resident, runs the fused packed-genome forward and
    returns softmax probabilities with the focal base appended as the last column."""

    REUSE_MIN_DENSITY = 0.1        # sites per base of a chunk's span above which the cross-position reuse path pays

    def __init__(self, model, fasta_path, local_radius, local_order=3, distal_radius=None, device="cuda", batch_sites=1 << 20,
                 model_type="snv", dirichlet_weights=None, poisson=None, scale_factor=None, reuse=True):
        """`dirichlet_weights` / `poisson` / `scale_factor`: apply the post-head calibration chain of run_predict.py:217-225
        (and scripts/scaling.py) on the device, fused behind the head (calibration.calibrate_device); the shard then carries
        float64 calibrated probabilities and the sink must not calibrate again.  `poisson=None` follows the reference's rule
        (run_predict.py:224: `poisson_calib or model_type == 'indel'`): on for indel models, off for snv.  `reuse`: let dense blocks of sites take the
        cross-position reuse kernels (same probabilities within rounding, tests/test_gpu_reuse.py)."""
        from .data import ingest
        if poisson is None:
            poisson = model_type == "indel"
        self.calibration = dict(dirichlet_weights=dirichlet_weights, poisson=poisson, scale_factor=scale_factor)
        self.calibrated = dirichlet_weights is not None or bool(poisson) or bool(scale_factor)
        self._ingest = ingest
        self.model = model.to(device).eval()
        self.fasta_path, self.device = fasta_path, torch.device(device)
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.local_radius, self.local_order, self.distal_radius = local_radius, local_order, distal_radius
        self.batch_sites, self.model_type, self.reuse = batch_sites, model_type, bool(reuse) and model_type == "snv"
        # the FASTA index (record offsets: one pass over the file) is built on a host thread -- the C++ scanner releases the GIL --
        # so the driver reads the BED beside it; `records` joins
        self._records, self._scan_error, self._early = None, None, None
        self._scan = threading.Thread(target=self._scan_fasta, daemon=True)
        self._scan.start()
        self._resident = (None, None)
        self._prefetch = None          # (chrom, thread, result box)
        self.seconds = {"pack_wait": 0.0, "pack": 0.0}
        self.reuse_sites = 0           # sites that went through the reuse kernels (diagnostics / tests)

    def _scan_fasta(self):
        try:
            self._records = {r.name: r for r in self._ingest.scan_fasta(self.fasta_path)}
        except Exception as e:      # noqa: BLE001  (re-raised by `records`)
            self._scan_error = e
            return
        # shards come in ascending chromosome-name order, so the first one is most likely the smallest name of the file: pack it
        # right away (still beside the driver's BED read); a BED without that chromosome just leaves the box unused
        if self._records:
            first = min(self._records)
            box = {}
            t0 = time.perf_counter()
            try:
                box["packed"] = self._ingest.pack_fasta_record(self.fasta_path, self._records[first])
            except Exception as e:      # noqa: BLE001  (re-raised by genome())
                box["error"] = e
            box["seconds"] = time.perf_counter() - t0
            self._early = (first, box)

    @property
    def records(self):
        self._scan.join()      # a finished thread joins at once; safe from the packer thread as well
        if self._scan_error is not None:
            raise self._scan_error
        return self._records

    # -- chromosome residency ---------------------------------------------------------------------------------------------
    def _pack(self, chrom, box):
        t0 = time.perf_counter()
        try:
            box["packed"] = self._ingest.pack_fasta_record(self.fasta_path, self.records[chrom])
        except Exception as e:      # noqa: BLE001  (re-raised by the consumer)
            box["error"] = e
        box["seconds"] = time.perf_counter() - t0

    def prefetch(self, chrom):
        """Start packing `chrom` on a host thread (the C++ packer releases the GIL); genome(chrom) picks the result up."""
        if chrom is None or chrom not in self.records or self._resident[0] == chrom:
            return
        if self._prefetch is not None and self._prefetch[0] == chrom:
            return
        if self._early is not None and self._early[0] == chrom:      # the scan thread already packed it
            return
        box = {}
        th = threading.Thread(target=self._pack, args=(chrom, box), daemon=True)
        th.start()
        self._prefetch = (chrom, th, box)

    def genome(self, chrom):
        if self._resident[0] != chrom:
            self._resident = (None, None)              # drop the previous chromosome before the next one is uploaded
            if chrom not in self.records:
                raise KeyError(chrom)                  # the reference's seq_records[chrom] lookup
            t0 = time.perf_counter()
            if self._prefetch is not None and self._prefetch[0] == chrom:
                _, th, box = self._prefetch
                th.join()
                self._prefetch = None
            elif self._early is not None and self._early[0] == chrom:      # packed by the scan thread (records joined it above)
                box, self._early = self._early[1], None
            else:
                box = {}
                self._pack(chrom, box)
            self.seconds["pack_wait"] += time.perf_counter() - t0
            self.seconds["pack"] += box.get("seconds", 0.0)
            if "error" in box:
                raise box["error"]
            packed, mask, n, amb = box["packed"]
            from .data.genome import PackedGenome
            self._resident = (chrom, PackedGenome(packed, mask, n, self.device, amb))
            self._early = None                         # an unused early pack is dropped with the first resident chromosome
        return self._resident[1]

    # -- compute ----------------------------------------------------------------------------------------------------------
    def _to_device(self, a, dtype):
        if isinstance(a, torch.Tensor):
            return a.to(self.device, dtype)
        return torch.from_numpy(np.ascontiguousarray(a)).to(self.device, dtype)

    @torch.no_grad()
    def __call__(self, chrom, pos, strand):
        """(rows, n_class + 1) for this rank's block of the chromosome's sites (numpy arrays or tensors, any device)."""
        g = self.genome(chrom)
        pos, strand = self._to_device(pos, torch.int64), self._to_device(strand, torch.uint8)
        n = pos.shape[0]
        k = self.model.n_class
        out = torch.empty((n, k + 1), dtype=torch.float64 if self.calibrated else torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            if self.model_type == "snv" and self.reuse and n > 0:
                logp, used = self.model.forward_packed_reuse(g, pos, strand, local_radius=self.local_radius, local_order=self.local_order,
                                                             min_density=self.REUSE_MIN_DENSITY, batch_sites=self.batch_sites,
                                                             return_reuse_count=True)
                self.reuse_sites += used
                self._finish(out, 0, logp, g, pos, strand)
                return out
            for r0 in range(0, n, self.batch_sites):
                p, st = pos[r0:r0 + self.batch_sites], strand[r0:r0 + self.batch_sites]
                if self.model_type == "snv":
                    logp = self.model.forward_packed(g, p, st, local_radius=self.local_radius, local_order=self.local_order)
                else:
                    logp = self.model.forward_packed(g, p, st, self.distal_radius)
                self._finish(out, r0, logp, g, p, st)
        return out

    def _finish(self, out, r0, logp, g, p, st):
        m = logp.shape[0]
        if self.model_type == "snv":
            out[r0:r0 + m, -1] = g.encode_kmer(p, st, 1, 1)[:, 1].to(out.dtype)     # complemented focal base 0..4
        else:
            out[r0:r0 + m, -1] = 0.0
        if self.calibrated:
            from .calibration import calibrate_device
            out[r0:r0 + m, :-1] = calibrate_device(logp, **self.calibration)
        else:
            out[r0:r0 + m, :-1] = torch.softmax(logp, dim=1)


# ------------------------------------------------------------------------------------------------------------------
# the prediction table
# ------------------------------------------------------------------------------------------------------------------
_NAME_STRIDE = 256


def _name_table(names):
    buf = C.create_string_buffer(max(len(names), 1) * _NAME_STRIDE)
    for i, nm in enumerate(names):
        raw = str(nm).encode()
        if len(raw) >= _NAME_STRIDE:
            raise ValueError(f"chromosome name longer than {_NAME_STRIDE - 1} bytes: {nm!r}")
        buf[i * _NAME_STRIDE:i * _NAME_STRIDE + len(raw)] = raw
    return buf


def _tsv_struct(names_buf, n_names, chrom_id, start, end, strand, label, prob, prob_f64, n_class, prob_stride, perm, n):
    t = _lib.MuralTsvRows()
    t.chrom_names, t.n_chroms, t.name_stride = C.cast(names_buf, C.c_char_p), max(n_names, 1), _NAME_STRIDE
    t.chrom_id, t.start, t.end, t.strand, t.label, t.prob, t.perm = chrom_id, start, end, strand, label, prob, perm
    t.prob_f64, t.n_class, t.prob_stride, t.n = int(prob_f64), int(n_class), int(prob_stride), int(n)
    return t


def format_rows_host(names, chrom_id, start, end, strand, label, prob, perm=None, threads=0):
    """Text of the prediction-table rows (no header) for host arrays, formatted by the C++ row formatter (csrc/tsv.hip): bytes.
    `names`: list of chromosome names, `chrom_id` indexes it (None = every row is names[0]); `strand` uint8 (1 = '-');
    `perm`: output row i = input row perm[i] (len(perm) rows are written: a sub-range of the sorted rows in the part-file mode)."""
    n = len(start)
    prob = np.asarray(prob)
    if prob.dtype not in (np.float32, np.float64):
        prob = prob.astype(np.float64)
    if prob.ndim != 2:
        prob = prob.reshape(n, -1)
    prob = np.ascontiguousarray(prob)
    cols = dict(start=np.ascontiguousarray(start, np.int64), end=np.ascontiguousarray(end, np.int64),
                strand=np.ascontiguousarray(strand, np.uint8), label=np.ascontiguousarray(label, np.float32))
    cid = None if chrom_id is None else np.ascontiguousarray(chrom_id, np.int32)
    pm = None if perm is None else np.ascontiguousarray(perm, np.int64)
    names_buf = _name_table(names)
    ptr = lambda a: None if a is None else a.ctypes.data     # noqa: E731
    t = _tsv_struct(names_buf, len(names), ptr(cid), ptr(cols["start"]), ptr(cols["end"]), ptr(cols["strand"]), ptr(cols["label"]),
                    ptr(prob) if prob.size else None, prob.dtype == np.float64, prob.shape[1], prob.shape[1], ptr(pm),
                    n if pm is None else len(pm))
    lib = _lib.lib()
    bound = int(lib.mural_tsv_row_bound(C.byref(t)))
    if bound < 0:
        _lib.check(_lib.MURAL_E_INVALID)
    out = np.empty(max(int(t.n) * bound, 1), np.uint8)
    nbytes = C.c_int64(0)
    _lib.check(lib.mural_tsv_format_host(C.byref(t), out.ctypes.data, out.size, C.byref(nbytes), int(threads)))
    return out[:nbytes.value].tobytes()


def _header(n_class):
    return ("\t".join(["chrom", "start", "end", "strand", "mut_type"] + ["prob%d" % i for i in range(n_class)]) + "\n").encode()


_PINNED_FREE = []      # pinned staging buffers of finished writers (pinning 100 MB costs ~20 ms: a process that writes many tables pays once)


def _pinned_staging(cap):
    for i, t in enumerate(_PINNED_FREE):
        if t.numel() >= cap:
            return _PINNED_FREE.pop(i)
    return torch.empty(cap, dtype=torch.uint8).pin_memory()


class _TextWriter(threading.Thread):
    """Writer thread of the device path: waits for a piece's format kernels, copies its text to pinned host memory on its own
    stream and write()s it; returns the device buffer to the pool.  One thread, pieces in order."""

    def __init__(self, fh, device, n_buffers, cap):
        super().__init__(daemon=True)
        self.fh, self.device = fh, device
        self.jobs, self.free = queue.Queue(), queue.Queue()
        self.text = [torch.empty(cap, dtype=torch.uint8, device=device) for _ in range(n_buffers)]
        self.count = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(n_buffers)]
        for i in range(n_buffers):
            self.free.put(i)
        self.host = _pinned_staging(cap)
        self.host_count = torch.zeros(1, dtype=torch.int64).pin_memory()
        self.error = None
        self.seconds = {"wait_device": 0.0, "copy": 0.0, "write": 0.0}
        self.bytes = 0
        self.shard_bytes = {}          # shard number -> bytes written for it (the part-file mode's index)

    def run(self):
        stream = torch.cuda.Stream(self.device)
        while True:
            job = self.jobs.get()
            if job is None:
                return
            idx, event, shard_no = job
            try:
                if self.error is None:
                    t0 = time.perf_counter()
                    event.synchronize()
                    t1 = time.perf_counter()
                    with torch.cuda.stream(stream):
                        self.host_count.copy_(self.count[idx], non_blocking=True)
                        stream.synchronize()
                        nb = int(self.host_count[0])
                        self.host[:nb].copy_(self.text[idx][:nb], non_blocking=True)
                        stream.synchronize()
                    t2 = time.perf_counter()
                    self.fh.write(memoryview(self.host.numpy())[:nb])
                    t3 = time.perf_counter()
                    self.seconds["wait_device"] += t1 - t0
                    self.seconds["copy"] += t2 - t1
                    self.seconds["write"] += t3 - t2
                    self.bytes += nb
                    self.shard_bytes[shard_no] = self.shard_bytes.get(shard_no, 0) + nb
            except Exception as e:      # noqa: BLE001  (surfaced by the sink)
                self.error = e
            finally:
                self.free.put(idx)


class TsvSink:
    """Rank-0 consumer of gathered shards: the prediction table of run_predict.py:217-239 (optional Dirichlet / Poisson
    calibration, columns chrom start end strand mut_type prob0.., rows sorted by (chrom, start), '%.4g'), byte-identical to the
    reference's pandas writer.

    A shard is a dict with the rows of ONE chromosome: ``chrom`` (name, or an array whose first entry is the name), ``start``,
    ``end``, ``strand`` (uint8 1 = '-', or '+' / '-' strings), ``label``, ``prob`` (n, k) -- numpy arrays, or torch tensors on a
    HIP device (then the stable sort by start, the calibration, the formatting and the copy-out all run on that device and a
    writer thread does the file I/O).  Shards whose chromosomes arrive in ascending name order -- what predict_bed_sharded
    produces -- are streamed out at once; any other arrival order is handled by spooling the raw rows and merging at close().
    Nothing is ever parsed back as numbers: chromosome names like '01' or '10' stay strings.

    ``parts=True`` under torch.distributed with more than one rank: EVERY rank is a consumer.  Each rank sorts the gathered shard (the
    all-gather of the probabilities stays the one collective of the path) and formats only ITS contiguous slice of the sorted rows --
    on its own GPU, through its own writer thread, into its own part file ``<path>.part<rank>`` -- so sort, format, copy-out and
    write() scale with the ranks instead of funnelling ~60 bytes of text per row through rank 0.  close() exchanges the per-shard byte
    counts and rank 0 strings the slices together in (shard, rank) order with in-kernel file copies (os.sendfile); the table is
    byte-identical to the single-writer one."""

    PIECE_ROWS = 1 << 20

    takes_aligned_blocks = True      # a shard marked "aligned" holds this rank's own rows in the table's order (no sort, no slicing)

    def __init__(self, path, poisson=False, dirichlet_weights=None, host_threads=0, parts=False, group=None):
        self.path, self.poisson, self.dirichlet_weights = str(path), poisson, dirichlet_weights
        self.host_threads = host_threads
        self.group = group
        self._emulated = isinstance(parts, tuple)      # (rank, world) without a process group: this rank's part file only -- the
        if self._emulated:                              # measurement of one rank's share of an N-rank run (bench.py: sink_only)
            self.rank, self.world = int(parts[0]), int(parts[1])
        else:
            self.rank = dist.get_rank(group) if (parts and dist.is_initialized()) else 0
            self.world = dist.get_world_size(group) if (parts and dist.is_initialized()) else 1
        self.parts = self.world > 1
        self._shard_no = -1
        self._shard_bytes = {}         # host path of the part mode: shard number -> bytes
        self._n_class = None
        self._out_path = self.path + (".part%04d" % self.rank if self.parts else "")
        self._fh = open(self._out_path, "wb")
        self._wrote_header = False
        self._last = None              # name of the last streamed chromosome
        self._spool = []               # out-of-order shards as host arrays
        self._writer = None
        self._ws = None
        self.rows = 0
        self.seconds = {"sort_format_enqueue": 0.0, "wait_buffer": 0.0, "host_format": 0.0, "host_write": 0.0}
        self._writer_totals = {"wait_device": 0.0, "copy": 0.0, "write": 0.0, "bytes": 0}
        self._writer_shard_bytes = {}  # device path of the part mode: shard number -> bytes, over every writer thread this sink had

    # -- helpers ----------------------------------------------------------------------------------------------------------
    @staticmethod
    def _name(shard):
        c = shard["chrom"]
        if isinstance(c, str):
            return c
        return str(c[0]) if len(c) else None

    @staticmethod
    def _strand_u8(s):
        if isinstance(s, torch.Tensor):
            return s
        s = np.asarray(s)
        return s if s.dtype == np.uint8 else (s == "-").astype(np.uint8)

    def _ensure_header(self, n_class):
        self._n_class = n_class
        if self.parts:                 # part files carry rows only; rank 0 writes the header when it strings them together
            return
        if not self._wrote_header:
            self._flush_writer()
            self._fh.write(_header(n_class))
            self._wrote_header = True

    def _flush_writer(self):
        """Wait until the writer thread has written everything handed to it (the file position is then ours)."""
        w = self._writer
        if w is not None:
            held = [w.free.get() for _ in range(len(w.text))]
            for i in held:
                w.free.put(i)
            if w.error is not None:
                raise w.error

    def _host_prob(self, prob):
        prob = np.asarray(prob)
        if self.dirichlet_weights is not None:
            from .calibration import dirichlet_calibrate
            prob = dirichlet_calibrate(prob, self.dirichlet_weights)
        if self.poisson:
            from .data.ingest import poisson_calibrate
            prob = poisson_calibrate(prob)
        return prob

    # -- streaming --------------------------------------------------------------------------------------------------------
    def _stream_host(self, name, shard):
        t0 = time.perf_counter()
        prob = self._host_prob(shard["prob"])
        start = np.asarray(shard["start"])
        if shard.get("aligned"):       # this rank's own rows, already in the table's order (predict_bed_sharded: _ShardTail.aligned)
            perm = np.arange(len(start), dtype=np.int64)
        else:
            perm = np.argsort(start, kind="stable")
            if self.parts:
                s0, s1 = shard_bounds(len(perm), self.rank, self.world)
                perm = perm[s0:s1]
        text = format_rows_host([name], None, start, shard["end"], self._strand_u8(shard["strand"]), shard["label"], prob, perm,
                                self.host_threads) if len(perm) else b""
        t1 = time.perf_counter()
        self._ensure_header(prob.shape[1] if prob.ndim == 2 else 0)
        self._flush_writer()
        self._fh.write(text)
        self._shard_bytes[self._shard_no] = self._shard_bytes.get(self._shard_no, 0) + len(text)
        self.seconds["host_format"] += t1 - t0
        self.seconds["host_write"] += time.perf_counter() - t1

    def _stream_device(self, name, shard):
        t0 = time.perf_counter()
        prob = shard["prob"]
        dev = prob.device
        n, k = prob.shape[0], int(shard.get("n_class", prob.shape[1]))
        lib = _lib.lib()
        with torch.cuda.device(dev):
            if self.dirichlet_weights is not None or self.poisson:
                from .calibration import calibrate_device
                prob = calibrate_device(prob[:, :k].contiguous() if prob.stride(0) != k else prob, dirichlet_weights=self.dirichlet_weights,
                                        poisson=self.poisson, input_is_prob=True)
            if prob.dtype not in (torch.float32, torch.float64):
                prob = prob.to(torch.float32)
            if prob.stride(1) != 1:
                prob = prob.contiguous()
            to = lambda a, dt: (a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a))).to(dev, dt).contiguous()   # noqa: E731
            start, end = to(shard["start"], torch.int64), to(shard["end"], torch.int64)
            strand, label = to(self._strand_u8(shard["strand"]), torch.uint8), to(shard["label"], torch.float32)
            if shard.get("aligned"):   # this rank's own rows, already in the table's order (predict_bed_sharded: _ShardTail.aligned)
                perm = torch.arange(n, dtype=torch.int64, device=dev)
            else:
                perm = torch.sort(start, stable=True).indices
                if self.parts:             # this rank's slice of the sorted rows
                    s0, s1 = shard_bounds(n, self.rank, self.world)
                    perm = perm[s0:s1].contiguous()
                    n = s1 - s0
            names_buf = _name_table([name])
            t = _tsv_struct(names_buf, 1, None, start.data_ptr(), end.data_ptr(), strand.data_ptr(), label.data_ptr(), prob.data_ptr(),
                            prob.dtype == torch.float64, k, prob.stride(0), None, 0)
            bound = int(lib.mural_tsv_row_bound(C.byref(t)))
            piece = self.PIECE_ROWS
            self._ensure_header(k)
            if self._writer is None or self._writer.device != dev or self._writer.text[0].numel() < piece * bound:
                self._close_writer()
                self._writer = _TextWriter(self._fh, dev, 3, piece * (bound + 24))     # slack: a longer chromosome name reuses it
                self._writer.start()
                self._ws = torch.empty(int(lib.mural_tsv_format_workspace_bytes(piece)) + _NAME_STRIDE, dtype=torch.uint8, device=dev)
            w = self._writer
            stream = _lib.current_stream_ptr(dev)
            for r0 in range(0, n, piece):
                m = min(piece, n - r0)
                tw = time.perf_counter()
                idx = w.free.get()                         # back-pressure: all text buffers are with the writer
                self.seconds["wait_buffer"] += time.perf_counter() - tw
                if w.error is not None:
                    w.free.put(idx)
                    raise w.error
                t.perm, t.n = perm[r0:r0 + m].data_ptr(), m
                _lib.check(lib.mural_tsv_format_device(C.byref(t), w.text[idx].data_ptr(), w.text[idx].numel(), w.count[idx].data_ptr(),
                                                      self._ws.data_ptr(), self._ws.numel(), stream))
                ev = torch.cuda.Event()
                ev.record()
                w.jobs.put((idx, ev, self._shard_no))
            # the tensors of this shard must outlive the kernels just enqueued: the caching allocator keeps their memory on this
            # stream, so later allocations of the same stream cannot overwrite them before the kernels ran
        self.seconds["sort_format_enqueue"] += time.perf_counter() - t0

    def __call__(self, shard):
        name = self._name(shard)
        n = len(shard["start"])
        # an aligned shard hands every rank ITS rows of a chromosome, in the table's order and possibly in several parts: the parts after
        # the first continue the shard (same name), and a rank without rows still counts the shard (the part files are strung together by
        # shard number)
        more = bool(shard.get("aligned")) and name is not None and name == self._last and not self._spool
        if name is not None and n == 0 and shard.get("aligned") and self.parts:
            if more:
                return
            if not (self._last is None or name > self._last):
                raise ValueError("TsvSink(parts=True) takes the shards in ascending chromosome order (predict_bed_sharded's order)")
            self._last = name
            self._shard_no += 1
            return
        if name is None or n == 0:
            return
        _refuse_second_calibration(shard, self.poisson, self.dirichlet_weights)
        self.rows += n
        on_device = isinstance(shard["prob"], torch.Tensor) and shard["prob"].is_cuda
        if self.parts and not more and not (self._last is None or name > self._last):
            raise ValueError("TsvSink(parts=True) takes the shards in ascending chromosome order (predict_bed_sharded's order)")
        if more or (not self._spool and (self._last is None or name > self._last)):
            if not more:
                self._last = name
                self._shard_no += 1
            if on_device:
                self._stream_device(name, shard)
            else:
                if isinstance(shard["prob"], torch.Tensor):
                    shard = dict(shard, prob=shard["prob"].numpy())
                self._stream_host(name, shard)
            return
        cpu = lambda a: a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)     # noqa: E731
        k = int(shard.get("n_class", shard["prob"].shape[1]))
        self._spool.append({"name": name, "start": cpu(shard["start"]), "end": cpu(shard["end"]),
                            "strand": cpu(self._strand_u8(shard["strand"])), "label": cpu(shard["label"]),
                            "prob": self._host_prob(cpu(shard["prob"])[:, :k])})

    # -- close ------------------------------------------------------------------------------------------------------------
    def _close_writer(self):
        if self._writer is not None:
            self._writer.jobs.put(None)
            self._writer.join()
            if len(_PINNED_FREE) < 2:
                _PINNED_FREE.append(self._writer.host)
            err = self._writer.error
            for key, v in dict(self._writer.seconds, bytes=self._writer.bytes).items():
                self._writer_totals[key] += v
            # per-shard byte counts of EVERY writer this sink has had (a later chromosome name longer than the writer's staging or
            # a device change replaces the writer mid-run): the part-file assembly's index
            for k, v in self._writer.shard_bytes.items():
                self._writer_shard_bytes[k] = self._writer_shard_bytes.get(k, 0) + v
            self._writer = None
            if err is not None:
                raise err

    def writer_seconds(self):
        """Busy seconds of the writer thread(s) by phase and the bytes they wrote (complete after close())."""
        out = dict(self._writer_totals)
        if self._writer is not None:
            for key, v in dict(self._writer.seconds, bytes=self._writer.bytes).items():
                out[key] += v
        return out

    @staticmethod
    def _copy_slices(dst_path, part_path, pieces):
        """Copy [(source offset, destination offset, bytes)] from this rank's part file into the table through a shared mapping: page
        faults of different ranks proceed side by side, where write()s to ONE file are serialised by its inode lock (the copy of a
        7 GB table by rank 0 alone took as long as the prediction)."""
        import mmap
        total = sum(n for _, _, n in pieces)
        if total == 0:
            return
        with open(dst_path, "r+b") as dst, open(part_path, "rb") as src:
            dm = mmap.mmap(dst.fileno(), 0)
            sm = mmap.mmap(src.fileno(), 0, access=mmap.ACCESS_READ)
            try:
                for so, do, n in pieces:
                    step = 64 << 20
                    for o in range(0, n, step):
                        m = min(step, n - o)
                        dm[do + o:do + o + m] = sm[so + o:so + o + m]
                dm.flush()
            finally:
                sm.close()
                dm.close()

    def _close_parts(self):
        """Part mode: exchange the per-shard byte counts; rank 0 creates the table at its final size (header + every slice), then EVERY
        rank copies its own slices to their places in (shard, rank) order -- in parallel -- and the part files go."""
        w_bytes = dict(self._shard_bytes)
        for k, v in self._writer_shard_bytes.items():
            w_bytes[k] = w_bytes.get(k, 0) + v
        mine = [int(w_bytes.get(i, 0)) for i in range(self._shard_no + 1)]
        self._fh.flush()
        self._fh.close()
        size = os.path.getsize(self._out_path)
        if sum(mine) != size:      # an index that does not add up to the part file would mis-order or truncate the table silently
            raise IOError("part file %s holds %d bytes, its per-shard index adds up to %d" % (self._out_path, size, sum(mine)))
        if self._emulated:
            return
        everyone = [None] * self.world
        dist.all_gather_object(everyone, (mine, self._n_class), group=self.group)     # (every rank closed its part before this returns)
        n_shards = max(len(c) for c, _ in everyone)
        k = next((nc for _, nc in everyone if nc is not None), 0)
        head = _header(k)
        count = lambda r, i: everyone[r][0][i] if i < len(everyone[r][0]) else 0      # noqa: E731
        t0 = time.perf_counter()
        pieces, dst_off, src_off = [], len(head), 0
        for i in range(n_shards):
            for r in range(self.world):
                n = count(r, i)
                if r == self.rank and n:
                    pieces.append((src_off, dst_off, n))
                    src_off += n
                dst_off += n
        ok = True
        if self.rank == 0:
            try:
                with open(self.path, "wb") as out:
                    out.write(head)
                    out.truncate(dst_off)
            except OSError:
                ok = False
        dist.barrier(group=self.group)      # the table file exists at its final size
        err = None
        try:
            if not ok:
                raise IOError("cannot create %s" % self.path)
            if src_off != size:
                raise IOError("part file %s: %d of %d bytes have a place in the table" % (self._out_path, src_off, size))
            self._copy_slices(self.path, self._out_path, pieces)
        except Exception as e:      # noqa: BLE001  (every rank reaches the barrier below; the parts stay for inspection)
            err = e
        flags = [None] * self.world
        dist.all_gather_object(flags, err is None, group=self.group)      # (also the barrier: every slice is in place)
        if all(flags):
            os.unlink(self._out_path)
        self.seconds["assemble_parts"] = time.perf_counter() - t0
        if err is not None:
            raise err
        if not all(flags):
            raise IOError("another rank could not copy its slices into %s; the part files stay" % self.path)

    def abort(self):
        """Stop the writer and remove what was written: the caller's run failed (e.g. the focal-base check of a later shard) and,
        like the reference, leaves no table behind."""
        try:
            self._close_writer()
        except Exception:      # noqa: BLE001  (the run is failing already)
            pass
        try:
            self._fh.close()
        finally:
            if os.path.exists(self._out_path):
                os.unlink(self._out_path)
        self._spool = []

    def close(self):
        self._close_writer()
        if self.parts:
            self._close_parts()
            return
        if not self._spool:
            if not self._wrote_header:
                self._fh.write(_header(0))
            self._fh.close()
            return
        # Out-of-order arrival: merge the spooled rows with what was already streamed.  Streamed rows are text already; per
        # chromosome they are merged line-wise by their start field (they arrived first, so they win ties), never re-parsed as
        # numbers.
        self._fh.flush()
        self._fh.close()
        with open(self.path, "rb") as fh:
            blob = fh.read()
        head_len = blob.index(b"\n") + 1 if self._wrote_header else 0
        k = self._spool[0]["prob"].shape[1]
        body = blob[head_len:]
        lines_by_chrom = {}
        if body:
            for line in body.split(b"\n")[:-1]:
                lines_by_chrom.setdefault(line.split(b"\t", 1)[0].decode(), []).append(line)
        spooled = {}
        for sh in self._spool:
            spooled.setdefault(sh["name"], []).append(sh)
        with open(self.path, "wb") as out:
            out.write(_header(k))
            for name in sorted(set(lines_by_chrom) | set(spooled)):
                old = lines_by_chrom.get(name, [])
                new_lines = []
                if name in spooled:
                    parts = spooled[name]
                    cat = lambda key: np.concatenate([p[key] for p in parts])     # noqa: E731
                    text = format_rows_host([name], None, cat("start"), cat("end"), cat("strand"), cat("label"), cat("prob"), None,
                                            self.host_threads)
                    new_lines = text.split(b"\n")[:-1]
                lines = old + new_lines
                starts = np.array([int(ln.split(b"\t", 2)[1]) for ln in lines], np.int64)
                for i in np.argsort(starts, kind="stable"):
                    out.write(lines[i] + b"\n")
        self._spool = []


def _refuse_second_calibration(rows, poisson, dirichlet_weights):
    """`rows` (a shard or a collected result) says whether its probabilities went through the calibration chain on the device
    already (HipShardForward(dirichlet_weights= / poisson= / scale_factor=); `poisson` defaults to ON there for indel models,
    run_predict.py:224).  Calibrating such rows again would be silent and wrong."""
    if rows is not None and rows.get("calibrated") and (poisson or dirichlet_weights is not None):
        raise ValueError("these probabilities are calibrated already (HipShardForward applied its dirichlet_weights / poisson / "
                         "scale_factor chain on the device; poisson defaults to on for model_type='indel'): drop poisson= / "
                         "dirichlet_weights= here, or build the forward with poisson=False")


def write_predictions(res, path, poisson=False, dirichlet_weights=None):
    """The prediction table of run_predict.py:217-239 from the dict returned by ``predict_bed`` / ``predict_bed_sharded``: optional
    Dirichlet calibration (``calibration.load_dirichlet_weights`` of the model's ``model.fdiri_cal.pkl``), optional Poisson
    calibration, then columns chrom, start, end, strand, mut_type, prob0.., rows sorted by (chrom, start) like pandas'
    sort_values (stable), tab-separated, floats as '%.4g' -- byte-identical to the reference's pandas writer, formatted by the
    C++ row formatter.  Returns the number of rows."""
    prob = np.asarray(res["prob"])
    _refuse_second_calibration(res, poisson, dirichlet_weights)
    if dirichlet_weights is not None:
        from .calibration import dirichlet_calibrate
        prob = dirichlet_calibrate(prob, dirichlet_weights)
    if poisson:
        from .data.ingest import poisson_calibrate
        prob = poisson_calibrate(prob)
    chrom = np.asarray(res["chrom"], dtype=object)
    n = len(chrom)
    names = sorted(set(chrom.tolist()))
    rank = {nm: i for i, nm in enumerate(names)}
    cid = np.fromiter((rank[c] for c in chrom), np.int32, n)
    start = np.asarray(res["start"], np.int64)
    perm = np.lexsort((start, cid))                     # stable, like pandas' multi-column sort_values
    strand = np.asarray(res["strand"])
    strand = strand if strand.dtype == np.uint8 else (strand == "-").astype(np.uint8)
    k = prob.shape[1] if prob.ndim == 2 else 0
    with open(path, "wb") as fh:
        fh.write(_header(k))
        if n:
            fh.write(format_rows_host(names, cid, start, res["end"], strand, np.asarray(res["label"], np.float32), prob, perm))
    return n


def _device_of(forward):
    dev = getattr(forward, "device", None)
    if dev is not None and torch.device(dev).type == "cuda":
        return torch.device(dev)
    return None


def _aligned_verdict(infos):
    """The focal-base check of an aligned shard from every rank's (rows, first segment, its '+' / '-' focal base, last segment, its
    '+' / '-' focal base, own verdict): a block's own groups were checked by its rank; a (segment, strand) group that runs over a block
    border -- or over several blocks, some of them without a row of that strand -- must carry one base.  Raises ValueError."""
    carry_seg, carry = None, [-1, -1]
    for m, seg_a, fa_p, fa_m, seg_b, fb_p, fb_m, bad in (tuple(int(v) for v in row) for row in infos):
        if bad:
            raise ValueError(_FOCAL_MSG)
        if m == 0:
            continue
        if carry_seg == seg_a:
            for have, mine in ((carry[0], fa_p), (carry[1], fa_m)):
                if have >= 0 and mine >= 0 and have != mine:
                    raise ValueError(_FOCAL_MSG)
        if seg_b != carry_seg:
            carry_seg, carry = seg_b, [-1, -1]
        # (seg_a == seg_b: the two triples describe the same groups)
        carry = [fb_p if fb_p >= 0 else carry[0], fb_m if fb_m >= 0 else carry[1]]


class _ShardTail:
    """What happens to a gathered shard (rows of one chromosome in bed_reader order): the per-(segment, strand) focal-base check --
    its verdict is read one shard late, so no rank waits for work just enqueued --, the sink, the collection for the caller."""

    def __init__(self, forward, model_type, sink, collect, T, dev, rank):
        self.model_type, self.sink, self.collect, self.T, self.dev = model_type, sink, collect, T, dev
        self.feeds_sink = sink is not None and (rank == 0 or getattr(sink, "parts", False))
        self.need_meta = collect or self.feeds_sink
        # HipShardForward with a calibration chain hands over calibrated probabilities: the flag travels with every shard (and the
        # collected result) so that a sink / write_predictions asked to calibrate as well refuses instead of calibrating twice
        self.calibrated = bool(getattr(forward, "calibrated", False))
        self.kept, self.pending = [], None
        T.update({"compute_enqueue": 0.0, "gather": 0.0, "sink": 0.0, "focal_wait": 0.0})

    def _finish_check(self):
        pc, self.pending = self.pending, None
        if pc is None:
            return
        t = time.perf_counter()
        ev, host = pc
        ev.synchronize()
        self.T["focal_wait"] += time.perf_counter() - t
        if host.dim() == 2:                 # an aligned shard: the ranks' border groups (see aligned())
            _aligned_verdict(host.numpy())
        elif int(host[0]) != 0:
            raise ValueError(_FOCAL_MSG)

    def __call__(self, chrom, runs, full, start, end, strand, label, grp, file_rows=None):
        """`full`: (n, k + 1) probabilities + focal base; `grp`: non-decreasing group ids; `runs`: [(lo, hi)] positions of the shard's
        rows in the whole input's bed_reader order; `file_rows`: file row index per row (rank-local ingest) or None."""
        n, k = full.shape[0], full.shape[1] - 1
        dev = self.dev
        if dev is not None:
            if self.model_type == "snv":
                status = torch.zeros(1, dtype=torch.int32, device=dev)
                with torch.cuda.device(dev):
                    _lib.check(_lib.lib().mural_focal_group_check(full.data_ptr(), int(full.dtype == torch.float64), full.stride(0), k,
                                                                 grp.contiguous().data_ptr(), n, status.data_ptr(),
                                                                 _lib.current_stream_ptr(dev)))
                    host = torch.zeros(1, dtype=torch.int32).pin_memory()
                    host.copy_(status, non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record()
                self._finish_check()               # the verdict of the PREVIOUS shard
                self.pending = (ev, host)
            shard = None
            if self.need_meta:
                shard = {"chrom": chrom, "start": start, "end": end, "strand": strand, "label": label, "prob": full, "n_class": k,
                         "calibrated": self.calibrated}
        else:
            full = full.cpu().numpy() if isinstance(full, torch.Tensor) else np.asarray(full)
            as_np = lambda a: a.numpy() if isinstance(a, torch.Tensor) else np.asarray(a)       # noqa: E731
            if self.model_type == "snv":
                check_focal_groups(full[:, -1].astype(np.int64), as_np(grp))
            shard = {"chrom": chrom, "start": as_np(start), "end": as_np(end), "strand": as_np(strand), "label": as_np(label),
                     "prob": full[:, :-1], "n_class": k, "calibrated": self.calibrated}
        if self.feeds_sink:
            t0 = time.perf_counter()
            self.sink(shard)
            self.T["sink"] += time.perf_counter() - t0
        if self.collect:
            cpu = lambda a: a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)     # noqa: E731
            self.kept.append((runs, {"start": cpu(shard["start"]), "end": cpu(shard["end"]), "strand": cpu(shard["strand"]),
                                     "label": cpu(shard["label"]), "prob": cpu(shard["prob"])[:, :k], "chrom": chrom,
                                     "file_rows": None if file_rows is None else cpu(file_rows)}))

    # -- a chromosome whose rows already are in the table's order (BedRun.in_order) -------------------------------------------------
    # The file order then IS the output order, this rank's block of the rows IS its slice of the table, and nothing but the focal-base
    # check looks across blocks: a (segment, strand) group that straddles a block border must agree on both sides.  Every rank checks its
    # own groups, the ranks exchange 7 numbers -- rows, first / last segment, focal base of the '+' and the '-' group of each (-1: none) --
    # and every rank walks the chain (verdict read one shard late, like the gathered shards').  No all-gather of rows, no sort of n rows.
    def aligned_part(self, chrom, local, start, end, strand, label, anchor, central_bp):
        """One PART of this rank's block of an aligned chromosome (the block goes through in parts of <= _ALIGNED_PART_ROWS rows, in file
        order: the table writer works on one part while the next is computed, and host / device memory is bounded by a part).  `local`:
        (m, k + 1) probabilities + focal base, start / strand / end / label: the part's site columns.  Returns the part's border record
        for ``aligned_close`` (None for models without the focal-base rule)."""
        m, k = local.shape[0], local.shape[1] - 1
        dev = self.dev
        info = None
        if self.model_type == "snv":
            e0 = (anchor if anchor is not None else 1) + central_bp
            if dev is not None:
                seg = torch.where(start > e0, (start - e0 + (central_bp - 1)) // central_bp, torch.zeros_like(start))
                key = (seg << 1) | strand.to(torch.int64)
                key_o, order = torch.sort(key, stable=True)
                focal_o = local[:, k][order].contiguous()
                status = torch.zeros(1, dtype=torch.int32, device=dev)
                info = torch.full((8,), -1, dtype=torch.int64, device=dev)
                info[0] = m
                if m:
                    with torch.cuda.device(dev):
                        _lib.check(_lib.lib().mural_focal_group_check(focal_o.data_ptr(), int(focal_o.dtype == torch.float64), 1, 0,
                                                                     key_o.data_ptr(), m, status.data_ptr(), _lib.current_stream_ptr(dev)))
                    want = torch.stack([key_o[0] & ~1, (key_o[0] & ~1) | 1, key_o[-1] & ~1, (key_o[-1] & ~1) | 1])
                    at = torch.searchsorted(key_o, want).clamp(max=m - 1)
                    have = key_o[at] == want
                    foc = torch.where(have, focal_o[at].to(torch.int64), torch.full_like(at, -1))
                    info[1], info[4] = key_o[0] >> 1, key_o[-1] >> 1
                    info[2:4], info[5:7] = foc[0:2], foc[2:4]
                info[7] = status[0].to(torch.int64)
            else:
                start_h, strand_h = np.asarray(start), np.asarray(strand)
                seg = np.where(start_h > e0, (start_h - e0 + (central_bp - 1)) // central_bp, 0)
                key = (seg.astype(np.int64) << 1) | strand_h.astype(np.int64)
                order = np.argsort(key, kind="stable")
                key_o = key[order]
                loc_h = local.cpu().numpy() if isinstance(local, torch.Tensor) else np.asarray(local)
                focal_o = loc_h[:, k][order].astype(np.int64)
                info = np.full(8, -1, np.int64)
                info[0], info[7] = m, 0
                if m:
                    try:
                        check_focal_groups(focal_o, key_o)
                    except ValueError:
                        info[7] = 1
                    want = np.array([key_o[0] & ~1, (key_o[0] & ~1) | 1, key_o[-1] & ~1, (key_o[-1] & ~1) | 1])
                    at = np.minimum(np.searchsorted(key_o, want), m - 1)
                    foc = np.where(key_o[at] == want, focal_o[at], -1)
                    info[1], info[4] = key_o[0] >> 1, key_o[-1] >> 1
                    info[2:4], info[5:7] = foc[0:2], foc[2:4]
                info = torch.from_numpy(info)
        shard = None
        if self.need_meta:
            as_np = (lambda a: a) if dev is not None else (lambda a: a.numpy() if isinstance(a, torch.Tensor) else np.asarray(a))     # noqa: E731
            prob = local if dev is not None else (local.cpu().numpy() if isinstance(local, torch.Tensor) else np.asarray(local))[:, :k]
            shard = {"chrom": chrom, "start": as_np(start), "end": as_np(end), "strand": as_np(strand), "label": as_np(label), "prob": prob,
                     "n_class": k, "calibrated": self.calibrated, "aligned": True}
        if self.feeds_sink:
            t0 = time.perf_counter()
            self.sink(shard)
            self.T["sink"] += time.perf_counter() - t0
        return info

    def aligned_close(self, infos, group, world, emulated):
        """The border records of this rank's parts of one aligned chromosome -> ONE small all-gather (parts x 8 numbers per rank) -> the
        chain over (rank, part) in table order; the verdict is read one shard late (no rank waits for work just enqueued)."""
        if self.model_type != "snv":
            return
        mine = torch.stack(infos)                                    # (parts, 8)
        if world > 1 and not emulated:
            every = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(every, mine, group=group)
            every = torch.cat(every)                                 # rank-major = table order
        else:
            every = mine
        if self.dev is not None:
            with torch.cuda.device(self.dev):
                host = torch.zeros(every.shape, dtype=torch.int64).pin_memory()
                host.copy_(every, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record()
            self._finish_check()               # the verdict of the PREVIOUS shard
            self.pending = (ev, host)
        else:
            self._finish_check()
            _aligned_verdict(every.numpy())

    def abort(self):
        # the verdict of a shard's focal-base check is read one shard late, i.e. after that shard's rows went to the sink: a failing
        # run must not leave a partial table with a valid-looking header behind (the reference exits before writing anything)
        if self.feeds_sink and hasattr(self.sink, "abort"):
            self.sink.abort()

    def close(self):
        self._finish_check()
        if self.feeds_sink and hasattr(self.sink, "close"):
            t0 = time.perf_counter()
            self.sink.close()
            self.T["sink_close"] = time.perf_counter() - t0

    def result(self, n_all, order):
        if not self.collect:
            return n_all
        if not self.kept:
            return {"chrom": np.zeros(0, object), "start": np.zeros(0, np.int64), "end": np.zeros(0, np.int64),
                    "strand": np.zeros(0, object), "label": np.zeros(0, np.float32), "prob": np.zeros((0, 0), np.float32),
                    "order": np.zeros(0, np.int64)}
        k = self.kept[0][1]["prob"].shape[1]
        if order is None:
            order = np.empty(n_all, np.int64)
        out = {"chrom": np.empty(n_all, object), "start": np.empty(n_all, np.int64), "end": np.empty(n_all, np.int64),
               "strand": np.empty(n_all, object), "label": np.empty(n_all, np.float32),
               "prob": np.empty((n_all, k), self.kept[0][1]["prob"].dtype), "order": order, "calibrated": self.calibrated}
        for runs, sh in self.kept:
            o = 0
            for lo, hi in runs:
                m = hi - lo
                out["chrom"][lo:hi] = sh["chrom"]
                out["start"][lo:hi], out["end"][lo:hi] = sh["start"][o:o + m], sh["end"][o:o + m]
                out["strand"][lo:hi] = np.where(sh["strand"][o:o + m] == 1, "-", "+")
                out["label"][lo:hi], out["prob"][lo:hi] = sh["label"][o:o + m], sh["prob"][o:o + m]
                if sh["file_rows"] is not None:
                    out["order"][lo:hi] = sh["file_rows"][o:o + m]
                o += m
        return out


def predict_bed_sharded(forward, bed_path, segment_center=300000, model_type="snv", group=None, sink=None, collect=True, timings=None,
                        ingest="ranked", emulate=None):
    """Sharded file-level prediction.  `forward(chrom_name, pos, strand) -> (rows, n_class + 1)` tensor (probabilities + focal
    base; see HipShardForward) is called once per shard (= chromosome, in ascending name order) with THIS rank's contiguous block
    of the shard's sites.  Every rank takes part in one all_gather per shard; `sink(shard_dict)` is called on rank 0 (every rank
    for a part-file sink) with the gathered shard in bed_reader order (keys chrom, start, end, strand, label, prob, n_class;
    device tensors when `forward` has a HIP ``device`` attribute, numpy arrays otherwise).  With `collect` the function also returns
    those arrays for ALL rows in bed_reader order on every rank -- leave it off for genome-scale inputs and let the sink stream
    them out.  Returns the dict (or the row count if not collecting).  `timings`: optional dict that receives the wall-clock split.

    `ingest="ranked"` (default): the BED file is indexed once -- every rank scans 1 / world of its bytes -- and a rank parses only
    its own block of every chromosome (``data.ingest.BedIndex``); the gathered row carries the site columns next to the
    probabilities, so host memory per rank is bounded by its share of one chromosome.  `ingest="whole"`: every rank parses the
    whole file (the reference's per-process BedTool, run_predict.py:107); the two produce identical results.
    `emulate=(rank, world)`: no process group -- run ONE rank's share of a `world`-rank run (its index scan, parse, compute,
    sort / format share with a part-file sink) and report the host seconds spent on standing in for the other ranks in
    timings['emulation'] (bench.py: config5_e2e.rank_share)."""
    if ingest == "whole":
        if emulate is not None:
            raise ValueError("emulate=(rank, world) needs ingest='ranked'")
        return _predict_bed_whole(forward, bed_path, segment_center, model_type, group, sink, collect, timings)
    if ingest != "ranked":
        raise ValueError(f"ingest must be 'ranked' or 'whole', got {ingest!r}")
    return _predict_bed_ranked(forward, bed_path, segment_center, model_type, group, sink, collect, timings, emulate)


def _row_layout(k, f64):
    """Byte layout of a gathered row: start i64 | end i64 | prob k x (f64 | f32) | focal (same type) | label f32 | strand u8 | pad."""
    e = 8 if f64 else 4
    off_prob = 16
    off_label = off_prob + (k + 1) * e
    off_strand = off_label + 4
    width = (off_strand + 1 + 7) // 8 * 8
    return off_prob, off_label, off_strand, width


def _pack_rows(local, start, end, strand, label):
    """(m, W) uint8 rows of this rank's block: the forward's (m, k + 1) matrix with the site columns beside it."""
    m, k = local.shape[0], local.shape[1] - 1
    f64 = local.dtype == torch.float64
    off_prob, off_label, off_strand, W = _row_layout(k, f64)
    buf = torch.zeros((m, W), dtype=torch.uint8, device=local.device)
    se = buf[:, 0:16].view(torch.int64)
    se[:, 0], se[:, 1] = start, end
    buf[:, off_prob:off_label].view(local.dtype)[:] = local
    buf[:, off_label:off_label + 4].view(torch.float32)[:, 0] = label
    buf[:, off_strand] = strand
    return buf


def _unpack_rows(full, k, dtype):
    off_prob, off_label, off_strand, _ = _row_layout(k, dtype == torch.float64)
    se = full[:, 0:16].view(torch.int64)
    return (full[:, off_prob:off_label].view(dtype), se[:, 0], se[:, 1], full[:, off_strand],
            full[:, off_label:off_label + 4].view(torch.float32)[:, 0])


def _bed_reader_keys(start, strand, run_rows, first_run_anchor, central_bp):
    """Sort key of bed_reader's row order (preprocessing.py:39-106) for the rows of one chromosome in FILE order: the runs restart
    the segment grid (end0 = 1 + central_bp; the file's very first run: its first start + central_bp), a row's segment is the
    number of times `while start > end0: end0 += central_bp` has fired so far -- a function of the running maximum of start --,
    and a segment yields its '+' rows, then its '-' rows: key = ((run << 40 | segment) << 1) | strand, rows stably sorted by it."""
    keys, lo = [], 0
    for j, rows in enumerate(run_rows):
        s = start[lo:lo + rows]
        e0 = (first_run_anchor if (j == 0 and first_run_anchor is not None) else 1) + central_bp
        rm = torch.cummax(s, 0).values
        seg = torch.where(rm > e0, (rm - e0 + (central_bp - 1)) // central_bp, torch.zeros_like(rm))
        keys.append((((j << 40) + seg) << 1) | strand[lo:lo + rows].to(torch.int64))
        lo += rows
    return keys[0] if len(keys) == 1 else torch.cat(keys)


_ALIGNED_BLOCKS = True      # (tests switch it off to compare the two routes of the ranked ingest)
_ALIGNED_PART_ROWS = 1 << 22


def _predict_bed_ranked(forward, bed_path, segment_center, model_type, group, sink, collect, timings, emulate):
    from .data import ingest
    if emulate is not None:
        rank, world = int(emulate[0]), int(emulate[1])
    else:
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        rank = dist.get_rank(group) if dist.is_initialized() else 0
    T = {} if timings is None else timings
    clock = time.perf_counter
    T["emulation"] = 0.0
    t0 = clock()
    index = ingest.BedIndex.build(bed_path, rank, world, group, emulate=emulate is not None, seconds=T)
    T["bed_index"] = clock() - t0
    if emulate is not None:
        T["emulation"] += T["bed_index"] - T["index_scan"]
    dev = _device_of(forward)
    tdev = dev if dev is not None else torch.device("cpu")
    tail = _ShardTail(forward, model_type, sink, collect, T, dev, rank)
    T.update({"bed_parse": 0.0, "pack_rows": 0.0, "reorder": 0.0})
    # aligned blocks need a consumer that takes a rank's own rows as its slice of the table: every rank's part-file sink (or the one
    # rank there is), and nobody who wants all rows back
    aligned_ok = (not collect and _ALIGNED_BLOCKS and (sink is None or getattr(sink, "takes_aligned_blocks", False))
                  and (world == 1 or sink is None or getattr(sink, "parts", False)))
    names = sorted(index.chroms)
    up = lambda a: torch.from_numpy(a).to(tdev)                                           # noqa: E731
    try:
        for si, chrom in enumerate(names):
            run_ids = index.chroms[chrom]
            run_rows = [index.runs[i].rows for i in run_ids]
            n = sum(run_rows)
            b0, b1 = shard_bounds(n, rank, world)
            # a chromosome whose rows already are in the table's order: this rank's block is its slice of the table (_ShardTail.aligned_part);
            # it goes through in parts (every rank the same number of them: the one collective of the shard carries a record per part)
            is_aligned = aligned_ok and len(run_ids) == 1 and index.runs[run_ids[0]].in_order
            n_parts = max(1, -(-(-(-n // world)) // _ALIGNED_PART_ROWS)) if is_aligned else 1
            records = []
            for part in range(n_parts):
                p0, p1 = shard_bounds(b1 - b0, part, n_parts)
                t0 = clock()
                start_h, end_h, label_h, strand_h = index.read_block(chrom, b0 + p0, b0 + p1)
                T["bed_parse"] += clock() - t0
                if part == 0 and hasattr(forward, "prefetch") and si + 1 < len(names):
                    forward.prefetch(names[si + 1])
                # every site column goes up BEFORE the forward is enqueued: a copy from pageable host memory waits for the stream's earlier
                # work, and behind the forward it would hold the host for the whole compute instead of letting it parse the next chromosome
                pos_b, strand_b, end_b, label_b = up(start_h), up(strand_h), up(end_h), up(label_h)
                t0 = clock()
                local = forward(chrom, pos_b if dev is not None else start_h, strand_b if dev is not None else strand_h)
                T["compute_enqueue"] += clock() - t0
                if local.shape[0] != p1 - p0:
                    raise RuntimeError("forward returned a wrong number of rows")
                if not isinstance(local, torch.Tensor):
                    local = torch.from_numpy(np.ascontiguousarray(local))
                if local.dtype not in (torch.float32, torch.float64):
                    local = local.to(torch.float32)
                if is_aligned:
                    anchor = index.runs[run_ids[0]].first_start if run_ids[0] == 0 else None
                    records.append(tail.aligned_part(chrom, local, pos_b, end_b, strand_b, label_b, anchor, int(segment_center)))
            if is_aligned:
                tail.aligned_close(records, group, world, emulate is not None)
                T["aligned_shards"] = T.get("aligned_shards", 0) + 1
                continue
            k = local.shape[1] - 1
            t0 = clock()
            packed = _pack_rows(local, pos_b, end_b, strand_b, label_b)
            T["pack_rows"] += clock() - t0
            t0 = clock()
            if emulate is not None and world > 1:
                # stand-in for the other ranks' blocks: their site columns (parsed here, outside the share) next to copies of this
                # rank's probability rows -- the gathered shard has the size, the sort keys and the text width of the real one
                if dev is not None:
                    torch.cuda.synchronize(dev)            # this rank's own work is charged to the share: the stand-in must not hide it
                te = clock()
                full = torch.empty((n, packed.shape[1]), dtype=torch.uint8, device=tdev)
                for r in range(world):
                    lo, hi = shard_bounds(n, r, world)
                    if r == rank or hi == lo:
                        continue
                    s2, e2, l2, d2 = index.read_block(chrom, lo, hi)
                    src = local[torch.arange(hi - lo, device=tdev) % max(local.shape[0], 1)] if local.shape[0] else \
                        torch.zeros((hi - lo, k + 1), dtype=local.dtype, device=tdev)
                    full[lo:hi] = _pack_rows(src, up(s2), up(e2), up(d2), up(l2))
                if dev is not None:
                    torch.cuda.synchronize(dev)            # ... and its device work is excluded with it
                T["emulation"] += clock() - te
                full[b0:b1] = packed                       # (the share's own copy: what the collective would deliver)
            else:
                full = all_gather_rows(packed, n, group)
            T["gather"] += clock() - t0
            t0 = clock()
            prob, start, end, strand, label = _unpack_rows(full, k, local.dtype)
            anchor = index.runs[run_ids[0]].first_start if run_ids[0] == 0 else None
            key = _bed_reader_keys(start, strand, run_rows, anchor, int(segment_center))
            key_o, order = torch.sort(key, stable=True)
            full_o = prob[order]                            # (n, k + 1) probabilities + focal base in bed_reader order
            start_o, strand_o = start[order], strand[order]
            end_o = label_o = file_rows = None
            if tail.need_meta:
                end_o, label_o = end[order], label[order]
            if collect:
                file_rows = torch.cat([index.runs[i].row0 + torch.arange(index.runs[i].rows, device=tdev) for i in run_ids])[order]
            T["reorder"] += clock() - t0
            runs = [(index.runs[i].row0, index.runs[i].row0 + index.runs[i].rows) for i in run_ids]
            tail(chrom, runs, full_o, start_o, end_o, strand_o, label_o, key_o, file_rows)
        tail.close()
    except BaseException:
        tail.abort()
        raise
    return tail.result(index.rows, None)


def _predict_bed_whole(forward, bed_path, segment_center, model_type, group, sink, collect, timings):
    """Every rank parses the whole BED (see predict_bed_sharded, ingest="whole")."""
    from .data import ingest
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    T = {} if timings is None else timings
    clock = time.perf_counter
    t0 = clock()
    sites = ingest.read_bed(bed_path)
    T["bed_read"] = clock() - t0
    t0 = clock()
    order, grp = ingest.bed_order(sites, segment_center)
    T["bed_order"] = clock() - t0
    n_all = len(order)
    dev = _device_of(forward)
    tail = _ShardTail(forward, model_type, sink, collect, T, dev, rank)
    t0 = clock()
    if dev is not None:
        # every column once to the device in FILE order; the bed_reader order is applied there (gathers at HBM speed)
        up = lambda a: torch.from_numpy(a).to(dev)                                            # noqa: E731
        order_d = up(order)
        cid_o = up(sites.chrom_id)[order_d]
        cut = torch.nonzero(cid_o[1:] != cid_o[:-1]).flatten().cpu().numpy() + 1 if n_all else np.zeros(0, np.int64)
        bounds = np.r_[0, cut, n_all] if n_all else np.zeros(1, np.int64)
        run_ids = cid_o[torch.from_numpy(bounds[:-1]).to(dev)].cpu().numpy() if n_all else np.zeros(0, np.int32)
        start_o, strand_o = up(sites.start)[order_d], up(sites.strand)[order_d]
        end_o = label_o = None
        if tail.need_meta:
            end_o, label_o = up(sites.end)[order_d], up(sites.score)[order_d]
        grp_o = up(grp)
        del cid_o
    else:
        cid_h = sites.chrom_id[order]
        runs = shard_runs(cid_h)
        bounds = np.array([lo for lo, _ in runs] + [n_all], np.int64)
        run_ids = np.array([cid_h[lo] for lo, _ in runs], np.int32)
        start_o, strand_o = sites.start[order], sites.strand[order]
        end_o, label_o = sites.end[order], sites.score[order]
        grp_o = grp
    T["order_columns"] = clock() - t0
    by_id = {}
    for i, c in enumerate(run_ids.tolist()):
        by_id.setdefault(c, []).append((int(bounds[i]), int(bounds[i + 1])))
    shards = [(sites.chrom_names[c], by_id[c]) for c in sorted(by_id, key=lambda c: sites.chrom_names[c])]

    def take(col, runs):
        if col is None:
            return None
        if len(runs) == 1:
            return col[runs[0][0]:runs[0][1]]
        parts = [col[lo:hi] for lo, hi in runs]
        return torch.cat(parts) if isinstance(col, torch.Tensor) else np.concatenate(parts)

    try:
        for si, (chrom, runs) in enumerate(shards):
            n = sum(hi - lo for lo, hi in runs)
            b0, b1 = shard_bounds(n, rank, world)
            pos_s, strand_s = take(start_o, runs), take(strand_o, runs)
            if hasattr(forward, "prefetch") and si + 1 < len(shards):
                forward.prefetch(shards[si + 1][0])
            t0 = clock()
            local = forward(chrom, pos_s[b0:b1], strand_s[b0:b1])
            T["compute_enqueue"] += clock() - t0
            if local.shape[0] != b1 - b0:
                raise RuntimeError("forward returned a wrong number of rows")
            t0 = clock()
            full = all_gather_rows(local, n, group)
            T["gather"] += clock() - t0
            tail(chrom, runs, full, pos_s, take(end_o, runs), strand_s, take(label_o, runs), take(grp_o, runs))
        tail.close()
    except BaseException:
        tail.abort()
        raise
    return tail.result(n_all, order)
