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
