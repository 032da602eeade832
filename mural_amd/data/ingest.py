"""FASTA / BED ingest for the packed-genome path (host side, C++ in csrc/ingest.hip) and a file-level predict helper.

Counterparts in the reference: ``SeqIO.to_dict(SeqIO.parse(ref_genome, 'fasta'))`` (MuRaL/data/preprocessing.py:836),
``bed_reader`` (:39-106), the per-batch focal-base consistency check (:479-484) and the prediction loop of
MuRaL/scripts/run_predict.py:188-239.  The reference spends >99 % of its wall time in per-character Python encoders
here; this path packs each chromosome once (2 bits per base + non-ACGT mask), keeps it in HBM and decodes windows inside
the kernels.  Rows come out in ``bed_reader`` order (per ``segment_center``-wide segment: '+' rows, then '-' rows), which
is the order the reference's DataLoader and its prediction table use before the final sort.
"""
import ctypes as C
import os
from dataclasses import dataclass

import numpy as np
import torch

from .. import _lib
from .genome import PackedGenome

_NAME = 256


@dataclass
class FastaRecord:
    name: str
    length: int
    offset: int


def scan_fasta(path):
    """List the records of a FASTA file without decoding them."""
    path = os.fspath(path)
    n = C.c_int64(0)
    _lib.check(_lib.lib().mural_fasta_scan(path.encode(), 0, _NAME, None, None, None, C.byref(n)))
    cnt = n.value
    names = C.create_string_buffer(max(cnt, 1) * _NAME)
    lengths = np.zeros(max(cnt, 1), np.int64)
    offsets = np.zeros(max(cnt, 1), np.int64)
    _lib.check(_lib.lib().mural_fasta_scan(path.encode(), cnt, _NAME, names, lengths.ctypes.data, offsets.ctypes.data,
                                          C.byref(n)))
    out = []
    for i in range(cnt):
        raw = names.raw[i * _NAME:(i + 1) * _NAME]
        out.append(FastaRecord(raw.split(b"\0", 1)[0].decode(), int(lengths[i]), int(offsets[i])))
    return out


def pack_fasta_record(path, rec):
    """(packed2 uint32[], nmask uint32[], length, (positions, symbols) of non-N ambiguity codes) of one record -- the same
    contract as ``genome.pack_sequence`` on the record's sequence string."""
    path = os.fspath(path)
    packed = np.zeros((rec.length + 15) // 16, np.uint32)
    mask = np.zeros((rec.length + 31) // 32, np.uint32)
    n_amb = C.c_int64(0)
    cap = 1024
    while True:
        amb, sym = np.zeros(cap, np.int64), np.zeros(cap, np.uint8)
        _lib.check(_lib.lib().mural_fasta_pack(path.encode(), rec.offset, rec.length, packed.ctypes.data, mask.ctypes.data,
                                              amb.ctypes.data, sym.ctypes.data, cap, C.byref(n_amb)))
        if n_amb.value <= cap:
            return packed, mask, rec.length, (amb[:n_amb.value].copy(), sym[:n_amb.value].copy())
        cap = int(n_amb.value)


def read_fasta(path, device="cuda", names=None):
    """{record id: PackedGenome on `device`} (only `names` if given), the packed counterpart of SeqIO.to_dict."""
    out = {}
    for rec in scan_fasta(path):
        if names is not None and rec.name not in names:
            continue
        if rec.name in out:
            raise ValueError(f"Duplicate key '{rec.name}'")        # SeqIO.to_dict raises ValueError on duplicate ids
        packed, mask, n, amb = pack_fasta_record(path, rec)
        out[rec.name] = PackedGenome(packed, mask, n, device, amb)
    return out


@dataclass
class BedSites:
    chrom_names: list
    chrom_id: np.ndarray    # int32, index into chrom_names
    start: np.ndarray       # int64, 0-based
    end: np.ndarray
    score: np.ndarray       # float32 class label (column 5)
    strand: np.ndarray      # uint8, 0 '+', 1 '-'

    def __len__(self):
        return len(self.start)


def read_bed(path):
    """Six-column BED (chrom start end name score strand) -> arrays in file order."""
    path = os.fspath(path)
    n, nc = C.c_int64(0), C.c_int32(0)
    lib = _lib.lib()
    _lib.check(lib.mural_bed_read(path.encode(), 0, None, None, None, None, None, 0, _NAME, None, C.byref(n), C.byref(nc)))
    rows, chroms = n.value, nc.value
    cid = np.zeros(rows, np.int32)
    start, end = np.zeros(rows, np.int64), np.zeros(rows, np.int64)
    score = np.zeros(rows, np.float32)
    strand = np.zeros(rows, np.uint8)
    names = C.create_string_buffer(max(chroms, 1) * _NAME)
    _lib.check(lib.mural_bed_read(path.encode(), rows, cid.ctypes.data, start.ctypes.data, end.ctypes.data, score.ctypes.data,
                                 strand.ctypes.data, chroms, _NAME, names, C.byref(n), C.byref(nc)))
    if n.value != rows or nc.value != chroms:      # (the file changed between the counting call and the fill call)
        raise RuntimeError(f"{path}: counted {rows} rows / {chroms} chromosomes, then read {n.value} / {nc.value}")
    cn = [names.raw[i * _NAME:(i + 1) * _NAME].split(b"\0", 1)[0].decode() for i in range(chroms)]
    return BedSites(cn, cid, start, end, score, strand)


PIECE_ROWS = 1 << 18      # rows per index piece at most: finding row r of a chromosome costs a newline scan of one piece


def _scan_bed_pieces(path, byte_lo, byte_hi, piece_rows):
    """([(name, byte_lo, byte_hi, rows, first_start, (in_order, last_start, first_strand, last_strand))], file_bytes) for the rows that
    start in [byte_lo, byte_hi)."""
    lib = _lib.lib()
    cap = 1 << 12
    while True:
        names = C.create_string_buffer(cap * _NAME)
        cols = [np.zeros(cap, np.int64) for _ in range(4)]
        order = np.zeros(4 * cap, np.int64)
        n, size = C.c_int64(0), C.c_int64(0)
        _lib.check(lib.mural_bed_index_scan(path.encode(), int(byte_lo), int(byte_hi), int(piece_rows), _NAME, cap, names,
                                           *(c.ctypes.data for c in cols), order.ctypes.data, C.byref(n), C.byref(size)))
        if n.value <= cap:
            break
        cap = int(n.value)
    raw = names.raw
    out = [(raw[i * _NAME:(i + 1) * _NAME].split(b"\0", 1)[0].decode(), int(cols[0][i]), int(cols[1][i]), int(cols[2][i]), int(cols[3][i]),
            tuple(int(v) for v in order[4 * i:4 * i + 4]))
           for i in range(n.value)]
    return out, int(size.value)


@dataclass
class BedRun:
    """Consecutive rows of one chromosome in file order (bed_reader restarts its segment grid at every such run)."""
    name: str
    row0: int               # file row index of the run's first row
    rows: int
    first_start: int        # start of the run's first row (the FILE's first run anchors its grid there, preprocessing.py:63-64)
    pieces: list            # [(byte_lo, byte_hi, rows)]
    in_order: bool = False  # every row's (start, strand) >= its predecessor's ('+' < '-'): the file order IS the output table's order
    last: tuple = (0, 0)    # (start, strand) of the run's last row


class BedIndex:
    """Chromosome runs of a BED file with byte offsets: what a rank needs to parse ONLY its own block of a chromosome's rows.

    The reference opens one BedTool per process over the whole file (MuRaL/scripts/run_predict.py:107) and advises to split big
    inputs by hand (MuRaL/commands/predict.py:134-137).  ``BedIndex.build`` scans 1 / world of the file's bytes on every rank
    (C++ host threads; first two fields of a row only) and exchanges the pieces found (a few KB: ``all_gather_object``, metadata
    only); ``read_rows`` then parses a row range of one chromosome.  gzip files are inflated once per process."""

    def __init__(self, path, pieces, file_bytes):
        self.path, self.file_bytes = os.fspath(path), file_bytes
        self.runs, self.chroms = [], {}
        row = 0
        for name, lo, hi, rows, first, (ordered, last_start, first_strand, last_strand) in pieces:
            if self.runs and self.runs[-1].name == name:
                run = self.runs[-1]
                run.pieces.append((lo, hi, rows))
                run.rows += rows
                run.in_order = run.in_order and bool(ordered) and (first, first_strand) >= run.last
            else:
                self.chroms.setdefault(name, []).append(len(self.runs))
                run = BedRun(name, row, rows, first, [(lo, hi, rows)], bool(ordered))
                self.runs.append(run)
            run.last = (last_start, last_strand)
            row += rows
        self.rows = row

    @classmethod
    def build(cls, path, rank=0, world=1, group=None, piece_rows=None, emulate=False, seconds=None):
        """`emulate`: no process group -- this process scans every rank's share itself and reports the time of ITS share
        (`rank` of `world`) in seconds['index_scan'] (the measurement of one rank's share of an N-rank run)."""
        import time
        path = os.fspath(path)
        piece_rows = PIECE_ROWS if piece_rows is None else piece_rows
        _, size = _scan_bed_pieces(path, 0, 0, piece_rows)
        cut = [size * r // world for r in range(world + 1)]
        t0 = time.perf_counter()
        mine, _ = _scan_bed_pieces(path, cut[rank], cut[rank + 1], piece_rows)
        if seconds is not None:
            seconds["index_scan"] = time.perf_counter() - t0
        if world == 1:
            everyone = [mine]
        elif emulate:
            everyone = [mine if r == rank else _scan_bed_pieces(path, cut[r], cut[r + 1], piece_rows)[0] for r in range(world)]
        else:
            import torch.distributed as dist
            everyone = [None] * world
            dist.all_gather_object(everyone, mine, group=group)
        return cls(path, [p for part in everyone for p in part], size)

    def chrom_rows(self, name):
        return sum(self.runs[i].rows for i in self.chroms[name])

    def read_rows(self, run_index, r0, r1):
        """(start int64, end int64, score float32, strand uint8) of rows r0 .. r1 - 1 of a run."""
        run = self.runs[run_index]
        if not (0 <= r0 <= r1 <= run.rows):
            raise ValueError(f"rows [{r0}, {r1}) outside a run of {run.rows} rows")
        n = r1 - r0
        start, end = np.zeros(n, np.int64), np.zeros(n, np.int64)
        score, strand = np.zeros(n, np.float32), np.zeros(n, np.uint8)
        if n:
            cum = 0
            first = last = None
            for i, (_, _, rows) in enumerate(run.pieces):
                if first is None and cum + rows > r0:
                    first, skip = i, r0 - cum
                if cum < r1:
                    last = i
                cum += rows
            _lib.check(_lib.lib().mural_bed_parse_range(self.path.encode(), run.pieces[first][0], run.pieces[last][1], skip, n,
                                                       run.name.encode(), start.ctypes.data, end.ctypes.data, score.ctypes.data,
                                                       strand.ctypes.data))
        return start, end, score, strand

    def read_block(self, name, b0, b1):
        """Rows b0 .. b1 - 1 of chromosome `name` in FILE order (its runs concatenated): (start, end, score, strand)."""
        parts, off = [], 0
        for i in self.chroms[name]:
            rows = self.runs[i].rows
            lo, hi = max(b0 - off, 0), min(b1 - off, rows)
            if lo < hi:
                parts.append(self.read_rows(i, lo, hi))
            off += rows
        if not parts:
            return np.zeros(0, np.int64), np.zeros(0, np.int64), np.zeros(0, np.float32), np.zeros(0, np.uint8)
        if len(parts) == 1:
            return parts[0]
        return tuple(np.concatenate([p[k] for p in parts]) for k in range(4))


def bed_order(sites, central_bp):
    """(order, group): bed_reader row order of `sites` (see batching.segment_order, here in C++)."""
    n = len(sites)
    order, group = np.zeros(n, np.int64), np.zeros(n, np.int64)
    ng = C.c_int64(0)
    _lib.check(_lib.lib().mural_bed_segment_order(sites.chrom_id.ctypes.data, sites.start.ctypes.data, sites.strand.ctypes.data,
                                                 n, int(central_bp), order.ctypes.data, group.ctypes.data, C.byref(ng)))
    return order, group


def predict_bed(model, fasta_path, bed_path, local_radius, local_order=3, distal_radius=None, segment_center=300000,
                device="cuda", batch_sites=1 << 20, model_type="snv"):
    """Predict every BED site from a FASTA + BED pair.  Returns a dict of arrays in bed_reader order: ``chrom``, ``start``,
    ``end``, ``strand`` ('+'/'-'), ``label`` and ``prob`` (n, n_class) = softmax of the model output (run_predict.py:214).

    SNV sites must share one focal base after strand complement within every (segment, strand) group, as the reference
    enforces for every SNV input (preprocessing.py:479-484 through the order-1 pass of prepare_local_data, :400; it exits,
    this raises ValueError).  This is the one-process case of ``mural_amd.predict.predict_bed_sharded``: chromosomes are
    streamed through the device one at a time."""
    from ..predict import HipShardForward, predict_bed_sharded
    # (poisson=False: this function returns the softmax itself for every model type -- the calibration chain of run_predict.py:217-225 is
    # write_predictions' / the sink's; HipShardForward alone would apply the Poisson step to indel models by default)
    fwd = HipShardForward(model, fasta_path, local_radius, local_order, distal_radius, device, batch_sites, model_type, poisson=False)
    return predict_bed_sharded(fwd, bed_path, segment_center, model_type)



def packed_segments(genomes, sites, order, group, local_radius, local_order=3, distal_radius=None, model_type="snv"):
    """The dataset segments of the reference's ``CombinedDatasetNP`` (MuRaL/data/preprocessing.py:937-944: item i = the rows of
    the i-th ``bed_reader`` group) encoded on the device from packed genomes: an iterator of
    ``(y (1, n, 1) float32, cont_x (1, n, 1) float64 zeros, cat_x (1, n, cols) int64, distal_x (1, n, 4, L) float32)``, the
    shapes its DataLoader(batch_size=1) hands to ``generate_data_batches``.  `genomes`: {chrom name: PackedGenome};
    `order`, `group`: from ``bed_order``."""
    cid, start, strand = sites.chrom_id[order], sites.start[order], sites.strand[order]
    label = sites.score[order]
    bounds = np.r_[0, np.nonzero(group[1:] != group[:-1])[0] + 1, len(group)] if len(group) else np.zeros(1, np.int64)
    for lo, hi in zip(bounds[:-1], bounds[1:]):
        g = genomes[sites.chrom_names[cid[lo]]]
        pos = torch.from_numpy(start[lo:hi]).to(g.device)
        st = torch.from_numpy(strand[lo:hi]).to(g.device)
        cat = g.encode_kmer(pos, st, local_radius, local_order, model_type)
        distal = g.encode_onehot(pos, st, distal_radius, model_type)
        y = torch.from_numpy(label[lo:hi].astype(np.float32)).reshape(1, -1, 1).to(g.device)
        yield y, torch.zeros((1, hi - lo, 1), dtype=torch.float64), cat.unsqueeze(0), distal.unsqueeze(0)


def train_batches_from_files(fasta_path, bed_path, batch_size, local_radius, local_order=3, distal_radius=None, segment_center=300000,
                             sampled_segments=10, shuffle=True, generator=None, model_type="snv", device="cuda"):
    """Training batches ``(y, cont_x, cat_x, distal_x)`` straight from a FASTA + BED pair in the order the reference's pipeline
    produces them (bed_reader segments -> CombinedDatasetNP items -> ``generate_data_batches`` with `sampled_segments`
    segments per group, preprocessing.py:1148-1226): C++ ingest, device-side window encoders, no per-base Python."""
    from .batching import generate_data_batches
    sites = read_bed(bed_path)
    order, group = bed_order(sites, segment_center)
    used = {sites.chrom_names[c] for c in np.unique(sites.chrom_id)}
    genomes = read_fasta(fasta_path, device, names=used)
    missing = used - set(genomes)
    if missing:
        raise KeyError(sorted(missing)[0])
    segs = packed_segments(genomes, sites, order, group, local_radius, local_order, distal_radius, model_type)
    return generate_data_batches(segs, sampled_segments, batch_size, shuffle=shuffle, generator=generator)


def poisson_calibrate(prob):
    """MuRaL/model/calibration.py:10-23 on an (n, n_class) array: lambda = -log(clip(prob0, 1e-10, 1)); the mutation
    classes become lambda * prob_k / (1 - prob0) and class 0 becomes 1 - lambda (applied for INDEL models and with
    --poisson_calib, run_predict.py:224-225)."""
    prob = np.asarray(prob)
    p0 = np.clip(prob[:, 0], 1e-10, 1.0)
    lam = -np.log(p0)
    out = prob.copy()
    with np.errstate(divide="ignore", invalid="ignore"):
        out[:, 1:] = lam[:, None] * prob[:, 1:] / (1 - p0)[:, None]
    out[:, 0] = 1 - lam
    return out


def write_predictions(res, path, poisson=False, dirichlet_weights=None):
    """The prediction table of run_predict.py:217-239 (see ``mural_amd.predict.write_predictions``, which formats it with the C++ row
    formatter of csrc/tsv.hip, byte-identical to the reference's pandas writer)."""
    from ..predict import write_predictions as impl
    return impl(res, path, poisson, dirichlet_weights)
