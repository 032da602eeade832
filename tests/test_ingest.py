"""Host-side ingest (csrc/ingest.hip): FASTA packer vs the numpy packer pinned by the encoder fixtures, BED reader and
bed_reader row order vs the reference's golden vectors (tests/golden/windowing.npz).  No GPU needed."""
import numpy as np
import pytest

from mural_amd.data import genome as G
from mural_amd.data import ingest as I
from tests import _util as U


def _wrap(seq, width):
    return "\n".join(seq[i:i + width] for i in range(0, len(seq), width))


@pytest.fixture()
def fasta(tmp_path):
    rng = np.random.default_rng(7)
    seqs = {}
    for name, n in (("chr1", 5003), ("chr2", 64), ("scaffold_3", 1)):
        s = "".join(rng.choice(list("ACGT"), size=n))
        seqs[name] = s
    s = list(seqs["chr1"])
    s[10:40] = "N" * 30
    s[100:130] = [c.lower() for c in s[100:130]]
    for i, ch in zip((200, 201, 333, 4000, 5002), "RYnKv"):
        s[i] = ch
    seqs["chr1"] = "".join(s)
    path = tmp_path / "g.fa"
    with open(path, "w") as fh:
        fh.write(">chr1 some description here\n" + _wrap(seqs["chr1"], 60) + "\n\n")
        fh.write(">chr2\r\n" + _wrap(seqs["chr2"], 17).replace("\n", "\r\n") + "\r\n")
        fh.write(">scaffold_3\tx\n" + seqs["scaffold_3"])          # no trailing newline
    return path, seqs


def test_fasta_scan_and_pack_match_numpy_packer(fasta):
    path, seqs = fasta
    recs = I.scan_fasta(path)
    assert [r.name for r in recs] == list(seqs)
    assert [r.length for r in recs] == [len(s) for s in seqs.values()]
    for rec in recs:
        packed, mask, n, amb = I.pack_fasta_record(path, rec)
        p2, m2, n2, a2 = G.pack_sequence(seqs[rec.name])
        assert n == n2
        assert np.array_equal(packed, p2) and np.array_equal(mask, m2)
        assert np.array_equal(amb[0], a2[0]) and np.array_equal(amb[1], a2[1])


def test_fasta_packer_threads_agree_with_numpy_packer_on_large_records(tmp_path):
    """Records large enough for several host threads (8 MB of text each): a chunk boundary inside a 32-base word, a record that ends
    inside a chunk, and a record whose ragged lines run past the bytes its first line's width predicts (serial tail)."""
    rng = np.random.default_rng(11)
    alphabet = np.frombuffer(b"ACGTNacgtRY", np.uint8)
    prob = [.23, .23, .23, .23, .02, .01, .01, .01, .01, .01, .01]
    a = rng.choice(alphabet, size=18_000_011, p=prob)
    b = rng.choice(alphabet, size=17_000_003, p=prob)
    path = tmp_path / "big.fa"
    with open(path, "wb") as fh:
        fh.write(b">A\n")
        whole = len(a) // 60 * 60
        fh.write(np.concatenate([a[:whole].reshape(-1, 60), np.full((whole // 60, 1), 10, np.uint8)], axis=1).tobytes())
        fh.write(a[whole:].tobytes() + b"\n>B ragged\n")
        fh.write(b[:200].tobytes() + b"\n")
        rest = b[200:]
        whole = len(rest) // 50 * 50
        fh.write(np.concatenate([rest[:whole].reshape(-1, 50), np.full((whole // 50, 1), 10, np.uint8)], axis=1).tobytes())
        fh.write(rest[whole:].tobytes() + b"\n>C\nACGTN\n")
    recs = I.scan_fasta(path)
    assert [(r.name, r.length) for r in recs] == [("A", len(a)), ("B", len(b)), ("C", 5)]
    for rec, seq in zip(recs, (a, b, np.frombuffer(b"ACGTN", np.uint8))):
        packed, mask, n, amb = I.pack_fasta_record(path, rec)
        p2, m2, n2, a2 = G.pack_sequence(seq.tobytes().decode())
        assert n == n2 and np.array_equal(packed, p2) and np.array_equal(mask, m2)
        assert np.array_equal(amb[0], a2[0]) and np.array_equal(amb[1], a2[1])


def test_fasta_rejects_non_nucleotide_characters(tmp_path):
    path = tmp_path / "bad.fa"
    path.write_text(">x\nACGTJACGT\n")
    rec = I.scan_fasta(path)[0]
    with pytest.raises(ValueError, match="not a nucleotide"):
        I.pack_fasta_record(path, rec)
    empty = tmp_path / "empty.fa"
    empty.write_text("")
    assert I.scan_fasta(empty) == []
    with pytest.raises(ValueError):
        I.scan_fasta(tmp_path / "missing.fa")


def test_bed_reader_and_segment_order_match_bed_reader_golden(tmp_path):
    fx = U.load("windowing.npz")
    path = tmp_path / "s.bed"
    with open(path, "w") as fh:
        fh.write("# comment\ntrack name=x\n")
        for c, s, st, sc in zip(fx["in_chrom"], fx["in_start"], fx["in_strand"], fx["in_score"]):
            fh.write(f"chr{c}\t{s}\t{s + 1}\t.\t{sc}\t{'-' if st else '+'}\n")
    sites = I.read_bed(path)
    assert len(sites) == len(fx["in_start"])
    assert np.array_equal(sites.start, fx["in_start"]) and np.array_equal(sites.end, fx["in_start"] + 1)
    assert np.array_equal(sites.strand, fx["in_strand"]) and np.array_equal(sites.score, fx["in_score"].astype(np.float32))
    names = np.asarray([int(n[3:]) for n in sites.chrom_names])
    assert np.array_equal(names[sites.chrom_id], fx["in_chrom"])
    order, group = I.bed_order(sites, int(fx["central"]))
    assert np.array_equal(names[sites.chrom_id][order], fx["out_chrom"])
    assert np.array_equal(sites.start[order], fx["out_start"])
    assert np.array_equal(sites.strand[order], fx["out_strand"])
    assert np.array_equal(group, fx["out_group"])


def test_bed_reader_errors(tmp_path):
    bad = tmp_path / "bad.bed"
    bad.write_text("chr1\t10\t11\t.\t0\n")
    with pytest.raises(ValueError, match="6 tab-separated"):
        I.read_bed(bad)
    bad.write_text("chr1\tten\t11\t.\t0\t+\n")
    with pytest.raises(ValueError, match="malformed"):
        I.read_bed(bad)
    empty = tmp_path / "e.bed"
    empty.write_text("")
    assert len(I.read_bed(empty)) == 0
    order, group = I.bed_order(I.read_bed(empty), 1000)
    assert len(order) == 0


def _messy_bed(path, seed=3, gz=False):
    """Several chromosome runs (chr2 twice), starts that jump back and forth inside a run (bed_reader's grid follows the running
    maximum), both strands, comment / blank lines in the middle; optionally gzip with two members (a bgzip-like file)."""
    import gzip
    rng = np.random.default_rng(seed)
    lines = ["# comment", "track name=x"]
    for chrom, n in (("chr2", 2500), ("chr1", 1800), ("chr10", 1), ("chr2", 700), ("scaffold_9", 40)):
        st = np.sort(rng.integers(0, 90000, size=n))
        jump = rng.random(n) < 0.03
        st = np.where(jump, rng.integers(0, 90000, size=n), st)
        for i, s_ in enumerate(st):
            lines.append(f"{chrom}\t{int(s_)}\t{int(s_) + 1 + int(rng.integers(0, 3))}\tsite{i}\t{int(rng.integers(0, 4))}\t{'+-'[int(rng.integers(0, 2))]}")
            if i % 977 == 5:
                lines += ["", "#mid comment"]
    text = ("\n".join(lines) + "\n").encode()
    if gz:
        half = text.rfind(b"\n", 0, len(text) // 2) + 1
        with open(path, "wb") as fh:
            fh.write(gzip.compress(text[:half]) + gzip.compress(text[half:]))
    else:
        with open(path, "wb") as fh:
            fh.write(text)
    return path


@pytest.mark.parametrize("gz", [False, True])
def test_bed_index_reads_any_block_of_a_chromosome(tmp_path, gz):
    """BedIndex (every rank scans 1 / world of the bytes, pieces of <= piece_rows rows) + read_block == the whole-file reader, for
    every world size, also on a multi-member gzip file."""
    path = _messy_bed(tmp_path / ("s.bed.gz" if gz else "s.bed"), gz=gz)
    whole = I.read_bed(path)
    chrom_of = np.asarray(whole.chrom_names)[whole.chrom_id]
    for world in (1, 2, 3, 8):
        idx = I.BedIndex.build(path, rank=world - 1, world=world, emulate=True, piece_rows=211)
        assert idx.rows == len(whole)
        assert [r.name for r in idx.runs] == ["chr2", "chr1", "chr10", "chr2", "scaffold_9"]
        assert idx.runs[0].first_start == whole.start[0]
        assert [r.row0 for r in idx.runs] == [0, 2500, 4300, 4301, 5001]
        for name in idx.chroms:
            sel = chrom_of == name
            n = idx.chrom_rows(name)
            assert n == int(sel.sum())
            for b0, b1 in {(0, n), (0, 0), (n // 3, 2 * n // 3), (max(n - 1, 0), n), (min(210, n), min(213, n)), (min(2499, n), min(2502, n))}:
                st, en, sc, sd = idx.read_block(name, b0, b1)
                assert np.array_equal(st, whole.start[sel][b0:b1]) and np.array_equal(en, whole.end[sel][b0:b1])
                assert np.array_equal(sc, whole.score[sel][b0:b1]) and np.array_equal(sd, whole.strand[sel][b0:b1])


def test_gzip_inputs_read_like_plain_ones(tmp_path, fasta):
    import gzip
    path, seqs = fasta
    gzp = tmp_path / "g.fa.gz"
    gzp.write_bytes(gzip.compress(path.read_bytes()))
    recs, recs_gz = I.scan_fasta(path), I.scan_fasta(gzp)
    assert [(r.name, r.length, r.offset) for r in recs] == [(r.name, r.length, r.offset) for r in recs_gz]
    for a, b in zip(recs, recs_gz):
        pa, pb = I.pack_fasta_record(path, a), I.pack_fasta_record(gzp, b)
        assert np.array_equal(pa[0], pb[0]) and np.array_equal(pa[1], pb[1]) and pa[2] == pb[2]
    plain = I.read_bed(_messy_bed(tmp_path / "s.bed"))
    packed = I.read_bed(_messy_bed(tmp_path / "s.bed.gz", gz=True))
    assert plain.chrom_names == packed.chrom_names
    for col in ("chrom_id", "start", "end", "score", "strand"):
        assert np.array_equal(getattr(plain, col), getattr(packed, col))
    bad = tmp_path / "bad.bed.gz"
    bad.write_bytes(gzip.compress(b"chr1\t1\t2\t.\t0\t+\n" * 1000)[:-20])      # truncated stream
    with pytest.raises(ValueError, match="zlib"):
        I.read_bed(bad)


def test_ranked_ingest_errors(tmp_path):
    bad = tmp_path / "bad.bed"
    bad.write_text("chr1\t10\t11\t.\t0\t+\nchr1\tten\t11\t.\t0\t+\n")
    with pytest.raises(ValueError, match="malformed"):
        I.BedIndex.build(bad)
    bad.write_text("chr1\t10\t11\t.\t0\t+\nchr1\t12\t13\t.\t0\n")
    idx = I.BedIndex.build(bad)                    # the index reads two fields; the block's parse reads all six
    with pytest.raises(ValueError, match="6 tab-separated"):
        idx.read_block("chr1", 0, 2)
    bad.write_text("chr1\t10\t11\t.\t0\t+\nchr1\t12\t13\t.\t0\t+\n")
    idx = I.BedIndex.build(bad)
    bad.write_text("chr1\t10\t11\t.\t0\t+\nchr7\t12\t13\t.\t0\t+\n")
    with pytest.raises(ValueError, match="changed behind the index"):
        idx.read_block("chr1", 0, 2)


def test_poisson_calibrate_and_prediction_table_match_reference(tmp_path):
    fx = U.load("output.npz")
    got = I.poisson_calibrate(fx["prob"])
    assert np.array_equal(np.isnan(got), np.isnan(fx["poisson"]))
    ok = ~np.isnan(got)
    assert np.array_equal(got[ok], fx["poisson"][ok])
    res = {"chrom": fx["chrom"].astype(object), "start": fx["start"], "end": fx["start"] + 1, "strand": fx["strand"].astype(object),
           "label": fx["label"].astype(np.float32), "prob": fx["prob"]}
    path = tmp_path / "pred.tsv"
    I.write_predictions(res, path)
    assert path.read_text() == str(fx["table"])


def test_mu_scaling_matches_reference_table(tmp_path):
    """apply_scaling against the table the reference's scripts/scaling.py:10-28 wrote from the golden prediction table (it reads
    the '%.4g' table back, scales, writes '%.4g' again); the factor formula of :76-93 on a hand example."""
    import io
    import pandas as pd
    from mural_amd.calibration import apply_scaling, mu_scaling_factor
    fx = U.load("output.npz")
    src = pd.read_csv(io.StringIO(str(fx["table"])), sep="\t")
    names = ["prob%d" % i for i in range(4)]
    got = apply_scaling(src[names].to_numpy(), float(fx["scale_factor"]))
    out = src.copy()
    out[names[1:]] = got[:, 1:]
    out["prob0"] = got[:, 0]
    buf = io.StringIO()
    out.to_csv(buf, sep="\t", index=False, float_format="%.4g")
    assert buf.getvalue() == str(fx["scaled_table"])
    prob = np.array([[0.9, 0.05, 0.03, 0.02], [0.8, 0.1, 0.05, 0.05]])
    assert mu_scaling_factor(prob, 1.2e-8, 0.25, 0.5) == pytest.approx(1.2e-8 * 2 * 0.25 / 0.5 / 0.3)
