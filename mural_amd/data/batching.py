"""Host-side row ordering and batching with the reference's contracts (pure index logic, no tensors moved).

  * ``segment_order``          -- row order produced by bed_reader (MuRaL/data/preprocessing.py:39-106): sites are cut
                                  into `central_bp`-wide segments along each chromosome and every segment yields its '+'
                                  rows, then its '-' rows.  Predictions come out in this order
                                  (run_predict.py:234) before the final sort by (chrom, start).
  * ``generate_data_batches``  -- two-level batching of MuRaL/data/preprocessing.py:1148-1226: `batch_segment` segments
                                  are concatenated, cut into `batch_size` batches, a short tail batch is PREPENDED to the
                                  next group, and the final short batch is emitted.  Works on any sequence of 4-tuples
                                  of tensors shaped like the reference's DataLoader output (leading dim 1).
"""
import numpy as np
import torch


def segment_order(chrom, start, strand, central_bp):
    """Return (order, group): `order` permutes the input rows (sorted by chrom/start, as a BED file is) into
    bed_reader order; `group` numbers the (segment, strand) groups the reference yields."""
    chrom = np.asarray(chrom)
    start = np.asarray(start, dtype=np.int64)
    strand = np.asarray(strand).astype(bool)
    order, group = [], []
    g = 0
    pos_rows, neg_rows = [], []

    def flush():
        nonlocal g, pos_rows, neg_rows
        if pos_rows:
            order.extend(pos_rows)
            group.extend([g] * len(pos_rows))
            g += 1
            pos_rows = []
        if neg_rows:
            order.extend(neg_rows)
            group.extend([g] * len(neg_rows))
            g += 1
            neg_rows = []

    cur_chrom, end0 = None, 0
    for i in range(len(start)):
        if cur_chrom is None:
            cur_chrom = chrom[i]
            end0 = int(start[i]) + central_bp          # the first segment starts at the first site (:62-64)
        if chrom[i] != cur_chrom:
            flush()
            cur_chrom = chrom[i]
            end0 = 1 + central_bp                      # later chromosomes start their grid at 1 (:77-78)
        if start[i] > end0:
            flush()
            while start[i] > end0:
                end0 += central_bp
        (neg_rows if strand[i] else pos_rows).append(i)
    flush()
    return np.asarray(order, dtype=np.int64), np.asarray(group, dtype=np.int64)


def _concat_segments(segs):
    y = torch.cat([s[0].squeeze(0) for s in segs])
    cat = torch.cat([s[2].squeeze(0) for s in segs])
    dist = torch.cat([s[3].squeeze(0) for s in segs])
    return y, cat, dist


def generate_data_batches(segment_loader, batch_segment, batch_size, shuffle=True, generator=None):
    """Yield (y, cont_x, cat_x, distal_x) batches; cont_x is float64 zeros (n, 1) like the reference (:1209)."""
    carry = None
    group = []

    def batches_of(y, cat, dist, last):
        nonlocal carry
        n = y.shape[0]
        idx = torch.randperm(n, generator=generator) if shuffle else torch.arange(n)
        for o in range(0, n, batch_size):
            sel = idx[o:o + batch_size]
            if sel.shape[0] < batch_size and not last:
                carry = (y[sel], cat[sel], dist[sel])
                return
            yield y[sel], torch.zeros((sel.shape[0], 1), dtype=torch.float64), cat[sel], dist[sel]

    it = iter(segment_loader)
    pending = next(it, None)
    while pending is not None:
        group.append(pending)
        pending = next(it, None)
        if len(group) >= batch_segment or pending is None:
            y, cat, dist = _concat_segments(group)
            group = []
            if carry is not None:                       # the short tail of the previous group goes FIRST (:1219-1225)
                y, cat, dist = torch.cat([carry[0], y]), torch.cat([carry[1], cat]), torch.cat([carry[2], dist])
                carry = None
            yield from batches_of(y, cat, dist, last=pending is None)
