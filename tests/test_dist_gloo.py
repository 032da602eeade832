"""world_size-2 gloo test (CPU) of the sharded-prediction host path: block partition + single padded all_gather."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from mural_amd.predict import predict_sites, shard_bounds
        pos = torch.arange(n, dtype=torch.int64) * 3 + 11
        strand = (torch.arange(n) % 2).to(torch.uint8)
        calls = []

        def fake_forward(p, s):                      # deterministic function of the site: stands in for the HIP model
            calls.append(p.shape[0])
            base = p.to(torch.float32).unsqueeze(1) * torch.tensor([1.0, 0.5, 0.25, 0.125])
            return base + s.to(torch.float32).unsqueeze(1)

        out = predict_sites(fake_forward, pos, strand)
        want = fake_forward(pos, strand)
        lo, hi = shard_bounds(n, rank, world)
        ok = torch.equal(out, want) and calls[0] == hi - lo
        # the strong-scaling shape of bench.py: the block in several timed slices, ONE gather at the end, then the sample check
        from mural_amd.predict import verify_gathered_rows
        calls.clear()
        out3 = predict_sites(fake_forward, pos, strand, steps=3)
        ok = ok and torch.equal(out3, want) and sum(calls) == hi - lo and len(calls) == 3
        diff, rows, owners = verify_gathered_rows(fake_forward, pos, strand, out3, sample=64)
        ok = ok and diff == 0.0 and rows == min(64, n) and (owners == world or n < 8)
        if n > 4:                                    # a gather that swapped two blocks must not pass the check
            bad = out3.clone()
            bad[0], bad[n - 1] = out3[n - 1], out3[0]
            ok = ok and verify_gathered_rows(fake_forward, pos, strand, bad, sample=n)[0] > 0
        q.put((rank, bool(ok), tuple(out.shape)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n", [0, 1, 5, 64, 1001])
def test_sharded_predict_world2(n):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res), res
    assert all(shape == (n, 4) for _, _, shape in res)


# ------------------------------------------------------------------------------------------------------------------
# file-level sharded driver (mural_amd.predict.predict_bed_sharded): per-chromosome shards, per-rank blocks, one gather per
# shard, rank-0 sink -- host logic on CPU with a stand-in for the HIP forward
# ------------------------------------------------------------------------------------------------------------------
def _write_bed(path, unsorted=False, mixed=False):
    import numpy as np
    rng = np.random.default_rng(5)
    rows = []
    for chrom, n in (("chr1", 157), ("chr10", 1), ("chr2", 64)):
        for p in np.sort(rng.choice(20000, size=n, replace=False)):
            rows.append((chrom, int(p), "+-"[int(p) % 2], int(p) % 4))
    if unsorted:                                    # chr1 rows come back after chr10: its rows form two shards
        rows = rows[:100] + rows[157:158] + rows[100:157] + rows[158:]
    with open(path, "w") as fh:
        for c, p, st, lab in rows:
            fh.write(f"{c}\t{p}\t{p + 1}\t.\t{lab}\t{st}\n")
    return rows


def _fake_shard_forward(calls, mixed_at=None):
    import numpy as np

    def fwd(chrom, pos, strand):
        calls.append((chrom, len(pos)))
        h = (pos.astype(np.float64) * 0.001 + len(chrom) + strand.astype(np.float64) * 0.5)
        logits = np.stack([np.sin(h), np.cos(h), np.sin(2 * h), np.cos(3 * h)], axis=1)
        prob = np.exp(logits) / np.exp(logits).sum(axis=1, keepdims=True)
        focal = np.zeros(len(pos))
        if mixed_at is not None:
            focal[pos == mixed_at] = 1
        return torch.from_numpy(np.concatenate([prob, focal[:, None]], axis=1).astype(np.float32))
    return fwd


def _bed_worker(rank, world, port, bed, out_path, unsorted, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import numpy as np
        from mural_amd.predict import TsvSink, predict_bed_sharded, shard_bounds
        calls = []
        sink = TsvSink(out_path) if rank == 0 else None
        res = predict_bed_sharded(_fake_shard_forward(calls), bed, segment_center=3000, sink=sink)
        solo = []
        want = _fake_shard_forward(solo)
        ok = True
        for chrom in np.unique(res["chrom"]):
            sel = res["chrom"] == chrom
            w = want(chrom, res["start"][sel], (res["strand"][sel] == "-").astype(np.uint8)).numpy()[:, :4]
            ok &= bool(np.array_equal(res["prob"][sel], w))
        # this rank evaluated only its block of every shard; one shard per chromosome, in ascending name order, also when the
        # chromosome's rows come as two runs of the BED (its runs are concatenated in bed_reader order)
        ok &= [c for c, _ in calls] == ["chr1", "chr10", "chr2"]
        sizes = [157, 1, 64]
        for (c, n), total in zip(calls, sizes):
            lo, hi = shard_bounds(total, rank, world)
            ok &= n == hi - lo
        q.put((rank, bool(ok), len(res["start"])))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("unsorted", [False, True])
def test_predict_bed_sharded_world2(tmp_path, unsorted):
    import numpy as np
    import pandas as pd
    from mural_amd.data.ingest import write_predictions
    from mural_amd.predict import predict_bed_sharded
    bed = str(tmp_path / "s.bed")
    rows = _write_bed(bed, unsorted=unsorted)
    out_path = str(tmp_path / "pred.tsv")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bed_worker, args=(r, 2, port, bed, out_path, unsorted, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res), res
    assert all(n == len(rows) for _, _, n in res)
    # the streamed table of rank 0 == the single-process table written in one go
    solo = predict_bed_sharded(_fake_shard_forward([]), bed, segment_center=3000)
    want_path = str(tmp_path / "want.tsv")
    write_predictions(solo, want_path)
    got, want = pd.read_csv(out_path, sep="\t"), pd.read_csv(want_path, sep="\t")
    assert list(got.columns) == list(want.columns) and len(got) == len(rows)
    assert got.equals(want)
    assert open(out_path).read() == open(want_path).read()


def test_predict_bed_sharded_focal_check_and_empty(tmp_path):
    from mural_amd.predict import TsvSink, predict_bed_sharded
    bed = str(tmp_path / "s.bed")
    rows = _write_bed(bed)
    with pytest.raises(ValueError, match="different bases"):
        predict_bed_sharded(_fake_shard_forward([], mixed_at=rows[30][1]), bed, segment_center=3000)
    empty = str(tmp_path / "empty.bed")
    open(empty, "w").close()
    sink = TsvSink(str(tmp_path / "e.tsv"))
    res = predict_bed_sharded(_fake_shard_forward([]), empty, sink=sink)
    assert len(res["start"]) == 0
    assert open(tmp_path / "e.tsv").read().startswith("chrom\tstart\tend\tstrand\tmut_type")


# ------------------------------------------------------------------------------------------------------------------
# part-file sink: every rank sorts the gathered shard, formats ITS slice of the sorted rows and writes its own part file; rank 0
# only strings the slices together (TsvSink(parts=True)) -- the N-rank file-to-file path without a rank-0 text funnel
# ------------------------------------------------------------------------------------------------------------------
def _parts_worker(rank, world, port, bed, out_path, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from mural_amd.predict import TsvSink, predict_bed_sharded
        sink = TsvSink(out_path, parts=True)
        ok = sink.parts and sink.world == world and sink.rank == rank
        T = {}
        n = predict_bed_sharded(_fake_shard_forward([]), bed, segment_center=3000, sink=sink, collect=False, timings=T)
        # every rank formatted about 1 / world of the rows, nobody all of them.  A chromosome whose rows are in the table's order reaches
        # the sink as this rank's own rows (an "aligned" shard), any other as the gathered shard
        ok = ok and not os.path.exists(out_path + ".part%04d" % rank)
        q.put((rank, bool(ok), (n, sink.rows, T.get("aligned_shards", 0))))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("unsorted", [False, True])
def test_part_file_sink_equals_single_writer_table(tmp_path, world, unsorted):
    from mural_amd.data.ingest import write_predictions
    from mural_amd.predict import predict_bed_sharded
    bed = str(tmp_path / "s.bed")
    rows = _write_bed(bed, unsorted=unsorted)
    out_path = str(tmp_path / "pred.tsv")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_parts_worker, args=(r, world, port, bed, out_path, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res), res
    assert all(n == len(rows) for _, _, (n, _, _) in res)
    if unsorted:      # chr1 comes as two runs: gathered by every rank; chr10 and chr2 are in order: every rank is handed its own rows
        assert all(al == 2 and handed >= 157 for _, _, (_, handed, al) in res)
        assert sum(handed for _, _, (_, handed, _) in res) == 157 * world + 1 + 64
    else:
        assert all(al == 3 for _, _, (_, _, al) in res) and sum(handed for _, _, (_, handed, _) in res) == len(rows)
    solo = predict_bed_sharded(_fake_shard_forward([]), bed, segment_center=3000)
    want_path = str(tmp_path / "want.tsv")
    write_predictions(solo, want_path)
    assert open(out_path, "rb").read() == open(want_path, "rb").read()
    assert sorted(os.listdir(tmp_path)) == ["pred.tsv", "s.bed", "want.tsv"]      # the part files are gone


# ------------------------------------------------------------------------------------------------------------------
# rank-local ingest (the default): every rank scans 1 / world of the BED's bytes, parses only its block of every chromosome and the
# gathered row carries the site columns -- the table is byte-identical to the one of the whole-file ingest, plain and gzip
# ------------------------------------------------------------------------------------------------------------------
def _ingest_worker(rank, world, port, bed, out_path, ingest, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import numpy as np
        from mural_amd.data import ingest as I
        from mural_amd.predict import TsvSink, predict_bed_sharded
        I.PIECE_ROWS = 97                                   # several pieces per chromosome and per rank
        parsed = []
        read_block = I.BedIndex.read_block
        I.BedIndex.read_block = lambda self, name, b0, b1: parsed.append(b1 - b0) or read_block(self, name, b0, b1)
        res = predict_bed_sharded(_fake_shard_forward([]), bed, segment_center=3000, sink=TsvSink(out_path, parts=True), ingest=ingest)
        ok = ingest == "whole" or sum(parsed) <= len(res["start"]) // world + 8      # a rank parsed its blocks and nothing else
        digest = float(np.sum(res["prob"].astype(np.float64) * (1 + np.arange(len(res["start"]))[:, None])))
        q.put((rank, bool(ok), (len(res["start"]), res["order"].tolist()[:50], digest)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("gz", [False, True])
def test_ranked_and_whole_file_ingest_give_the_same_table(tmp_path, world, gz):
    from tests.test_ingest import _messy_bed
    bed = str(_messy_bed(tmp_path / ("s.bed.gz" if gz else "s.bed"), gz=gz))
    tables, results = {}, {}
    for ingest in ("ranked", "whole"):
        out_path = str(tmp_path / f"pred_{ingest}.tsv")
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_ingest_worker, args=(r, world, port, bed, out_path, ingest, q)) for r in range(world)]
        for p in procs:
            p.start()
        res = [q.get(timeout=120) for _ in procs]
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
        assert all(ok for _, ok, _ in res), res
        assert len({repr(r[2]) for r in res}) == 1          # every rank collected the same rows
        results[ingest] = res[0][2]
        tables[ingest] = open(out_path, "rb").read()
    assert results["ranked"] == results["whole"]             # row count, bed_reader order, probabilities
    assert tables["ranked"] == tables["whole"] and tables["ranked"].count(b"\n") == 5041 + 1


def test_ranked_ingest_single_process_equals_whole_and_emulates_a_rank(tmp_path):
    import numpy as np
    from mural_amd.data import ingest as I
    from mural_amd.predict import TsvSink, predict_bed_sharded
    from tests.test_ingest import _messy_bed
    bed = str(_messy_bed(tmp_path / "s.bed"))
    old = I.PIECE_ROWS
    I.PIECE_ROWS = 131
    try:
        a = predict_bed_sharded(_fake_shard_forward([]), bed, segment_center=3000, ingest="ranked")
        b = predict_bed_sharded(_fake_shard_forward([]), bed, segment_center=3000, ingest="whole")
        for key in ("chrom", "start", "end", "strand", "label", "prob", "order"):
            assert np.array_equal(a[key], b[key]), key
        # one rank's share of a 4-rank run, no process group: its own block is computed, the others are stood in for
        calls, T = [], {}
        sink = TsvSink(str(tmp_path / "p.tsv"), parts=(2, 4))
        n = predict_bed_sharded(_fake_shard_forward(calls), bed, segment_center=3000, sink=sink, collect=False, emulate=(2, 4), timings=T)
        assert n == len(a["start"]) and T["emulation"] > 0 and T["bed_parse"] > 0
        assert sum(m for _, m in calls) in range(n // 4 - 4, n // 4 + 5)
        with pytest.raises(ValueError, match="ranked"):
            predict_bed_sharded(_fake_shard_forward([]), bed, ingest="whole", emulate=(0, 2))
    finally:
        I.PIECE_ROWS = old


def test_failed_run_leaves_no_table_behind(tmp_path):
    """The focal-base verdict of a shard arrives after its rows went to the sink: the failing run removes the partial table (the
    reference exits before it writes anything, preprocessing.py:482-484)."""
    from mural_amd.predict import TsvSink, predict_bed_sharded
    bed = str(tmp_path / "s.bed")
    rows = _write_bed(bed)
    out_path = str(tmp_path / "pred.tsv")
    sink = TsvSink(out_path)
    with pytest.raises(ValueError, match="different bases"):
        predict_bed_sharded(_fake_shard_forward([], mixed_at=rows[200][1]), bed, segment_center=3000, sink=sink)
    assert not os.path.exists(out_path)


# ------------------------------------------------------------------------------------------------------------------
# aligned shards: a chromosome whose rows already are in the table's order is never gathered -- a rank's block is its slice of the
# table -- and only the focal-base check looks across block borders
# ------------------------------------------------------------------------------------------------------------------
def _ordered_bed(path, conflict=None):
    """Rows in the table's order with everything the aligned route has to get right: both strands at one start ('+' row first), long
    (segment, strand) groups that run over block borders, a chromosome shorter than the number of ranks, one out of order (general
    route in the same run).  `conflict` = (chrom, row): that row's focal base differs from its group's."""
    import numpy as np
    rng = np.random.default_rng(11)
    rows = []
    for chrom, n, span in (("chrA", 400, 9000), ("chrB", 2, 100), ("chrC", 301, 2500), ("chrD", 90, 50000)):
        pos = np.sort(rng.choice(span, size=n, replace=False))
        for i, p in enumerate(pos):
            rows.append((chrom, int(p), "+", 0))
            if i % 3 == 0:
                rows.append((chrom, int(p), "-", 1))      # the same start on the other strand: '+' first, as the table has it
    d = [r for r in rows if r[0] == "chrD"]
    rows = [r for r in rows if r[0] != "chrD"] + d[40:] + d[:40]      # chrD: one run, but not in order
    with open(path, "w") as fh:
        for c, p, st, lab in rows:
            fh.write(f"{c}\t{p}\t{p + 1}\t.\t{lab}\t{st}\n")
    return rows


def _ordered_forward(conflict=None):
    import numpy as np
    base = _fake_shard_forward([])

    def fwd(chrom, pos, strand):
        out = base(chrom, pos, strand).numpy()
        out[:, 4] = strand          # focal base: one per strand, so every (segment, strand) group agrees ...
        if conflict is not None and chrom == conflict[0]:
            out[(pos == conflict[1]) & (strand == conflict[2]), 4] = 3      # ... but for one row
        return torch.from_numpy(out)
    return fwd


def _aligned_worker(rank, world, port, bed, out_path, aligned, conflict, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from mural_amd import predict as P
        from mural_amd.data import ingest as I
        I.PIECE_ROWS = 53
        P._ALIGNED_BLOCKS = aligned
        P._ALIGNED_PART_ROWS = 37          # a rank's block of chrA / chrC goes through in several parts (chrB: fewer rows than ranks)
        T = {}
        sink = P.TsvSink(out_path, parts=True)
        try:
            n = P.predict_bed_sharded(_ordered_forward(conflict), bed, segment_center=700, sink=sink, collect=False, timings=T)
            q.put((rank, "ok", (n, sink.rows, T.get("aligned_shards", 0))))
        except ValueError as e:
            q.put((rank, "ValueError", str(e)))
    finally:
        dist.destroy_process_group()


def _run_aligned(tmp_path, world, bed, name, aligned, conflict=None):
    out_path = str(tmp_path / name)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_aligned_worker, args=(r, world, port, bed, out_path, aligned, conflict, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return out_path, res


@pytest.mark.parametrize("world", [2, 3, 5])
def test_aligned_shards_give_the_gathered_route_s_table(tmp_path, world):
    from mural_amd.data.ingest import write_predictions
    from mural_amd.predict import predict_bed_sharded
    bed = str(tmp_path / "o.bed")
    rows = _ordered_bed(bed)
    a_path, a = _run_aligned(tmp_path, world, bed, "aligned.tsv", True)
    g_path, g = _run_aligned(tmp_path, world, bed, "gathered.tsv", False)
    assert all(kind == "ok" for _, kind, _ in a + g), (a, g)
    assert all(al == 3 for _, _, (_, _, al) in a) and all(al == 0 for _, _, (_, _, al) in g)       # chrA, chrB, chrC; chrD is gathered
    n_d = sum(r[0] == "chrD" for r in rows)
    assert sum(handed for _, _, (_, handed, _) in a) == len(rows) - n_d + world * n_d            # own rows, and the gathered chrD everywhere
    assert all(handed == len(rows) for _, _, (_, handed, _) in g)
    solo = predict_bed_sharded(_ordered_forward(), bed, segment_center=700, ingest="whole")
    want_path = str(tmp_path / "want.tsv")
    write_predictions(solo, want_path)
    want = open(want_path, "rb").read()
    assert open(a_path, "rb").read() == want and open(g_path, "rb").read() == want and want.count(b"\n") == len(rows) + 1


@pytest.mark.parametrize("world", [2, 3])
def test_aligned_shards_find_a_focal_base_conflict_across_block_borders(tmp_path, world):
    """A (segment, strand) group of an aligned chromosome runs over several blocks: one odd row anywhere in it fails the run on EVERY rank,
    whichever rank holds it -- also when it is the only row of its group inside its block."""
    bed = str(tmp_path / "o.bed")
    rows = _ordered_bed(bed)
    a_rows = [r for r in rows if r[0] == "chrA"]
    for pick in (0, len(a_rows) // world - 1, len(a_rows) // world, len(a_rows) - 1):      # first row, last of block 0, first of block 1, last row
        c, p, st, _ = a_rows[pick]
        out_path, res = _run_aligned(tmp_path, world, bed, "t%d.tsv" % pick, True, conflict=(c, p, 1 if st == "-" else 0))
        assert all(kind == "ValueError" and "different bases" in msg for _, kind, msg in res), (pick, res)
        assert not os.path.exists(out_path) and not any(f.startswith("t%d.tsv.part" % pick) for f in os.listdir(tmp_path))


def test_aligned_shards_in_parts_single_process(tmp_path):
    """One process: the aligned route in parts of 29 rows writes the table of the gathered route, single-writer sink and collect=False;
    a conflict in the middle of a group that runs over parts is found."""
    from mural_amd import predict as P
    bed = str(tmp_path / "o.bed")
    rows = _ordered_bed(bed)
    old = P._ALIGNED_BLOCKS, P._ALIGNED_PART_ROWS
    try:
        tables = {}
        for aligned in (True, False):
            P._ALIGNED_BLOCKS, P._ALIGNED_PART_ROWS = aligned, 29
            T = {}
            out = str(tmp_path / ("t%d.tsv" % aligned))
            n = P.predict_bed_sharded(_ordered_forward(), bed, segment_center=700, sink=P.TsvSink(out), collect=False, timings=T)
            assert n == len(rows) and T.get("aligned_shards", 0) == (3 if aligned else 0)
            tables[aligned] = open(out, "rb").read()
        assert tables[True] == tables[False] and tables[True].count(b"\n") == len(rows) + 1
        P._ALIGNED_BLOCKS = True
        c, p, st, _ = [r for r in rows if r[0] == "chrC"][150]
        out = str(tmp_path / "bad.tsv")
        with pytest.raises(ValueError, match="different bases"):
            P.predict_bed_sharded(_ordered_forward((c, p, 1 if st == "-" else 0)), bed, segment_center=700, sink=P.TsvSink(out), collect=False)
        assert not os.path.exists(out)
    finally:
        P._ALIGNED_BLOCKS, P._ALIGNED_PART_ROWS = old


def test_aligned_verdict_walks_groups_over_block_borders():
    """The chain logic alone: rows of (rows, first segment, its '+' / '-' base, last segment, its '+' / '-' base, own verdict) per rank."""
    import numpy as np
    from mural_amd.predict import _aligned_verdict
    ok = lambda *rows: _aligned_verdict(np.array(rows, np.int64))          # noqa: E731

    def bad(*rows):
        with pytest.raises(ValueError, match="different bases"):
            _aligned_verdict(np.array(rows, np.int64))

    ok((5, 0, 1, 2, 3, 1, 2, 0), (4, 3, 1, 2, 7, 0, 3, 0))                  # segment 3 continues with the same bases; 7 is new
    bad((5, 0, 1, 2, 3, 1, 2, 0), (4, 3, 0, 2, 7, 0, 3, 0))                 # '+' of segment 3 changes at the border
    bad((5, 0, 1, 2, 3, 1, 2, 0), (4, 3, 1, 0, 7, 0, 3, 0))                 # '-' of segment 3 changes at the border
    ok((5, 0, 1, 2, 3, 1, 2, 0), (4, 4, 0, 0, 7, 0, 3, 0))                  # another segment: nothing to compare
    # the middle rank lies inside segment 3 and has '+' rows only: the '-' group is carried over it
    ok((5, 0, 1, 2, 3, 1, 2, 0), (2, 3, 1, -1, 3, 1, -1, 0), (4, 3, 1, 2, 9, 1, 1, 0))
    bad((5, 0, 1, 2, 3, 1, 2, 0), (2, 3, 1, -1, 3, 1, -1, 0), (4, 3, 1, 0, 9, 1, 1, 0))
    # a rank without rows in between changes nothing
    bad((5, 0, 1, 2, 3, 1, 2, 0), (0, -1, -1, -1, -1, -1, -1, 0), (4, 3, 0, 2, 9, 1, 1, 0))
    ok((5, 0, 1, 2, 3, 1, 2, 0), (0, -1, -1, -1, -1, -1, -1, 0), (4, 3, 1, 2, 9, 1, 1, 0))
    # a group first seen on a later rank of the same segment: the first rank had no '-' row of segment 3
    ok((5, 0, 1, 2, 3, 1, -1, 0), (4, 3, 1, 2, 3, 1, 2, 0), (1, 3, -1, 2, 3, -1, 2, 0))
    bad((5, 0, 1, 2, 3, 1, -1, 0), (4, 3, 1, 2, 3, 1, 2, 0), (1, 3, -1, 0, 3, -1, 0, 0))
    bad((5, 0, 1, 2, 3, 1, 2, 0), (4, 5, 1, 2, 7, 0, 3, 1))                 # a rank's own verdict


def _overlap_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from mural_amd.predict import OverlappedGather
        rows, width, steps = 7, 3, 6
        og = OverlappedGather(rows, width, torch.float32, torch.device("cpu"))
        seen = []
        for s in range(steps):
            local = torch.full((rows, width), float(100 * s + rank)) + torch.arange(rows)[:, None]
            i = og.submit(local)
            seen.append((s, i))
        og.finish()
        # the two buffers hold the last two steps, every rank's block in rank order
        ok = True
        for s, i in seen[-2:]:
            want = torch.cat([torch.full((rows, width), float(100 * s + r)) + torch.arange(rows)[:, None] for r in range(world)])
            ok &= bool(torch.equal(og.result(i), want))
        ok &= [i for _, i in seen] == [0, 1, 0, 1, 0, 1]
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_overlapped_gather_double_buffers_the_step_collective(world):
    """bench.py's weak-scaling collective: asynchronous all-gathers into two alternating buffers deliver every rank's rows in rank order."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_overlap_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok in res), res
