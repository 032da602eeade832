"""GPU parity for BASELINE config 5 (whole-genome SNV predict with the shipped Homo_sapiens weights, sharded): the same
driver the 8-rank run uses, here at world size 1 with the REAL HIP forward -- site-level (ShardedPredictor / predict_sites)
and file-level (predict_bed_sharded + HipShardForward + TsvSink) -- against the oracle model fed by the oracle encoders.
The multi-rank host logic (block partition, one gather per shard, sink) is covered on CPU by tests/test_dist_gloo.py."""
import numpy as np
import pytest
import torch

from oracle import encode_ref
from tests import _util as U

pytestmark = pytest.mark.gpu

PROB_TOL = 1e-5
HUMAN = ["snv_pretrained_human_AT.npz", "snv_pretrained_human_CpG.npz", "snv_pretrained_human_nonCpG.npz"]


def _models(name):
    from tests.test_gpu_snv import product_from_hp
    fx = U.load(name)
    model, _ = product_from_hp(fx["hp"])
    orc = U.snv_oracle_from_hp(fx["hp"])
    sd = U.snv_state_for(fx, orc)
    model.load_state_dict(sd)
    orc.load_state_dict(sd)
    orc.eval()
    return model.cuda().eval(), orc, int(fx["hp"][0]), int(fx["hp"][2])


def _genome(rng, n, iupac=True):
    alphabet = b"ACGTNRYKV" if iupac else b"ACGTN"
    p = [.2465, .2465, .2465, .2465, .01, .001, .001, .001, .001] if iupac else [.2475, .2475, .2475, .2475, .01]
    return rng.choice(np.frombuffer(alphabet, np.uint8), size=n, p=p).tobytes().decode()


def _oracle_probs(orc, seq, pos, neg, r, R):
    codes = encode_ref.seq_to_codes(seq)
    sym = ["-" if v else "+" for v in neg]
    cat = torch.from_numpy(encode_ref.kmer_encode(codes, pos, sym, r, 3))
    x = torch.from_numpy(encode_ref.onehot_encode(codes, pos, sym, R))
    with torch.no_grad():
        return torch.softmax(orc((torch.zeros(len(pos), 1, dtype=torch.float64), cat), x), dim=1).numpy()


@pytest.mark.parametrize("name", HUMAN)
def test_sharded_predictor_real_forward_matches_oracle(name):
    from mural_amd.data import PackedGenome
    from mural_amd.predict import ShardedPredictor
    model, orc, r, R = _models(name)
    rng = np.random.default_rng(55)
    seq = _genome(rng, 30_000)
    pos = np.sort(rng.choice(len(seq), size=300, replace=False))
    neg = rng.integers(0, 2, size=len(pos)).astype(np.uint8)
    sp = ShardedPredictor(model, PackedGenome.from_sequence(seq, "cuda"), local_radius=r, local_order=3)
    got = torch.softmax(sp(torch.from_numpy(pos).cuda(), torch.from_numpy(neg).cuda()), dim=1).cpu().numpy()
    assert np.abs(got - _oracle_probs(orc, seq, pos, neg, r, R)).max() <= PROB_TOL


def test_file_level_sharded_driver_streams_chromosomes(tmp_path):
    """FASTA + BED -> per-chromosome shards -> sorted prediction table, with IUPAC codes in the genome."""
    import pandas as pd
    from mural_amd.data.ingest import write_predictions
    from mural_amd.predict import HipShardForward, TsvSink, predict_bed_sharded
    model, orc, r, R = _models(HUMAN[0])
    rng = np.random.default_rng(56)
    seqs = {"chr1": _genome(rng, 9000), "chr2": _genome(rng, 6000), "chrX": _genome(rng, 4000)}
    fa = tmp_path / "g.fa"
    fa.write_text("".join(f">{k}\n" + "\n".join(s[i:i + 60] for i in range(0, len(s), 60)) + "\n" for k, s in seqs.items()))
    rows = []
    for name, s in seqs.items():                        # A on '+', T on '-': one focal base after complement
        arr = np.frombuffer(s.encode(), np.uint8)
        for p in np.sort(rng.choice(len(s), size=250, replace=False)):
            if arr[p] == ord("A"):
                rows.append((name, int(p), "+"))
            elif arr[p] == ord("T"):
                rows.append((name, int(p), "-"))
    bed = tmp_path / "s.bed"
    bed.write_text("".join(f"{c}\t{p}\t{p + 1}\t.\t{i % 4}\t{st}\n" for i, (c, p, st) in enumerate(rows)))
    fwd = HipShardForward(model, fa, local_radius=r, local_order=3, batch_sites=64)
    loaded = []
    inner = fwd.genome
    fwd.genome = lambda chrom: (loaded.append(chrom), inner(chrom))[1]
    out = tmp_path / "pred.tsv"
    res = predict_bed_sharded(fwd, bed, segment_center=2000, sink=TsvSink(out))
    assert [c for i, c in enumerate(loaded) if i == 0 or loaded[i - 1] != c] == ["chr1", "chr2", "chrX"]
    assert fwd._resident[0] == "chrX"                   # exactly one chromosome is resident at a time
    assert len(res["start"]) == len(rows)
    for name, s in seqs.items():
        sel = res["chrom"] == name
        want = _oracle_probs(orc, s, res["start"][sel], res["strand"][sel] == "-", r, R)
        assert np.abs(res["prob"][sel] - want).max() <= PROB_TOL, name
    want_path = tmp_path / "want.tsv"
    write_predictions(res, want_path)
    assert open(out).read() == open(want_path).read()
    df = pd.read_csv(out, sep="\t")
    assert df[["chrom", "start"]].equals(df[["chrom", "start"]].sort_values(["chrom", "start"]).reset_index(drop=True))


# ------------------------------------------------------------------------------------------------------------------
# SURVEY.md section 8f-2 on the device: softmax -> full-Dirichlet map -> Poisson -> mu scaling in one kernel
# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("tag", ["snv", "indel"])
def test_device_dirichlet_map_matches_reference_calibrator(tag):
    """G12: outputs of the reference's own FullDirichletCalibrator.predict_proba on two shipped calibrators."""
    from mural_amd.calibration import calibrate_device
    fx = U.load("dirichlet.npz")
    prob = torch.from_numpy(fx[tag + "_prob"]).cuda()
    got = calibrate_device(prob, dirichlet_weights=fx[tag + "_w"], input_is_prob=True)
    assert got.dtype == torch.float64 and got.is_cuda
    assert np.abs(got.cpu().numpy() - fx[tag + "_cal"]).max() <= 1e-6      # float32 log on the device vs numpy: last-bit differences
    # from the model output (log-probabilities): softmax happens inside the kernel
    got2 = calibrate_device(torch.log(prob), dirichlet_weights=fx[tag + "_w"]).cpu().numpy()
    assert np.abs(got2 - fx[tag + "_cal"]).max() <= 1e-5


def test_device_poisson_and_scaling_match_reference_tables(tmp_path):
    """G11: poisson_calibrate and apply_scaling outputs written by the reference (prob -> poisson -> '%.4g' table -> scaled)."""
    import io
    import pandas as pd
    from mural_amd.calibration import apply_scaling, calibrate_device
    from mural_amd.data.ingest import poisson_calibrate
    fx = U.load("output.npz")
    prob = torch.from_numpy(fx["prob"]).cuda()
    got = calibrate_device(prob, poisson=True, input_is_prob=True).cpu().numpy()
    # (row 0 of the fixture has prob0 = 1: the reference's 0 / 0 = NaN for the mutation classes, reproduced)
    assert np.isnan(fx["poisson"]).any() and np.array_equal(np.isnan(got), np.isnan(fx["poisson"]))
    np.testing.assert_allclose(got, fx["poisson"], rtol=3e-7, atol=1e-9, equal_nan=True)
    np.testing.assert_allclose(got, poisson_calibrate(fx["prob"].astype(np.float64)), rtol=1e-12, atol=1e-15, equal_nan=True)
    factor = float(fx["scale_factor"])
    scaled = calibrate_device(prob, poisson=True, scale_factor=factor, input_is_prob=True).cpu().numpy()
    np.testing.assert_allclose(scaled, apply_scaling(poisson_calibrate(fx["prob"].astype(np.float64)), factor), rtol=1e-12,
                               atol=1e-15, equal_nan=True)
    # the reference's scaling script (scripts/scaling.py:10-28) read the '%.4g' table of the raw probabilities: same resolution
    ref = pd.read_csv(io.StringIO(str(fx["scaled_table"])), sep="\t")
    src = pd.read_csv(io.StringIO(str(fx["table"])), sep="\t")
    cols = [c for c in ref.columns if c.startswith("prob")]
    plain = calibrate_device(torch.from_numpy(src[cols].to_numpy().astype(np.float32)).cuda(), scale_factor=factor,
                             input_is_prob=True).cpu().numpy()
    assert np.allclose(plain, ref[cols].to_numpy(), rtol=1e-3, atol=1e-9)
    # other class counts take the same kernel (2: template, 3: generic loop)
    for k in (2, 3, 8):
        p = torch.softmax(torch.randn(1000, k, generator=torch.Generator().manual_seed(k)), dim=1)
        w = np.random.default_rng(k).normal(size=(k, k + 1))
        from mural_amd.calibration import dirichlet_calibrate
        want = poisson_calibrate(dirichlet_calibrate(p.numpy(), w))
        got = calibrate_device(p.cuda(), dirichlet_weights=w, poisson=True, input_is_prob=True).cpu().numpy()
        assert np.abs(got - want).max() <= 1e-6 * max(1.0, np.abs(want).max())
    assert calibrate_device(torch.zeros(0, 4).cuda(), poisson=True).shape == (0, 4)


def test_sharded_driver_with_device_calibration_writes_the_same_table(tmp_path):
    """HipShardForward with the calibration fused behind the head + a plain sink == host-side calibration in the sink."""
    from mural_amd.predict import HipShardForward, TsvSink, predict_bed_sharded
    model, orc, r, R = _models(HUMAN[1])
    rng = np.random.default_rng(57)
    seq = _genome(rng, 8000, iupac=False)
    fa = tmp_path / "g.fa"
    fa.write_text(">chr7\n" + seq + "\n")
    arr = np.frombuffer(seq.encode(), np.uint8)
    sites = [int(p) for p in np.sort(rng.choice(len(seq), size=400, replace=False)) if arr[p] == ord("C")]
    bed = tmp_path / "s.bed"
    bed.write_text("".join(f"chr7\t{p}\t{p + 1}\t.\t0\t+\n" for p in sites))
    w = np.random.default_rng(58).normal(size=(4, 5)) * 0.2 + np.hstack([np.eye(4), np.zeros((4, 1))])
    a, b = tmp_path / "dev.tsv", tmp_path / "host.tsv"
    predict_bed_sharded(HipShardForward(model, fa, r, 3, dirichlet_weights=w, poisson=True), bed, sink=TsvSink(a), collect=False)
    predict_bed_sharded(HipShardForward(model, fa, r, 3), bed, sink=TsvSink(b, poisson=True, dirichlet_weights=w), collect=False)
    ta, tb = open(a).read().split("\n"), open(b).read().split("\n")
    assert len(ta) == len(tb) == len(sites) + 2
    diff = sum(x != y for x, y in zip(ta, tb))
    assert diff <= len(sites) // 100        # '%.4g' of float64 values that differ in the 7th digit: a rare rounding flip at most


def test_sharded_driver_takes_the_reuse_path_on_dense_sites_and_matches_oracle(tmp_path):
    """dense site lists go through the cross-position reuse kernels inside the sharded file-level driver (device
    columns, device sort, device-formatted table); probabilities against the oracle, table against the single-process writer.  The
    BED interleaves its chromosomes, so one of them arrives as two runs, and a third chromosome is sparse (per-window kernels)."""
    from mural_amd.data.ingest import write_predictions
    from mural_amd.predict import HipShardForward, TsvSink, predict_bed_sharded
    model, orc, r, R = _models(HUMAN[0])
    rng = np.random.default_rng(61)
    seqs = {"chr2": _genome(rng, 12_000), "chr10": _genome(rng, 9000), "chrS": _genome(rng, 30_000, iupac=False)}
    fa = tmp_path / "g.fa"
    fa.write_text("".join(f">{k}\n" + "\n".join(s[i:i + 70] for i in range(0, len(s), 70)) + "\n" for k, s in seqs.items()))

    def at_sites(name, lo, hi, every=1):
        arr = np.frombuffer(seqs[name].encode(), np.uint8)
        return [(name, int(p), "+" if arr[p] == ord("A") else "-") for p in range(lo, hi, every) if arr[p] in (ord("A"), ord("T"))]

    a, b, c = at_sites("chr2", 0, 6000), at_sites("chr10", 0, 9000), at_sites("chr2", 6000, 12_000)
    sparse = at_sites("chrS", 0, 30_000, every=41)
    rows = a + b + c + sparse                              # chr2 comes back after chr10: two runs of one chromosome
    bed = tmp_path / "s.bed"
    bed.write_text("".join(f"{ch}\t{p}\t{p + 1}\t.\t{i % 4}\t{st}\n" for i, (ch, p, st) in enumerate(rows)))
    fwd = HipShardForward(model, fa, local_radius=r, local_order=3)
    out = tmp_path / "pred.tsv"
    timings = {}
    res = predict_bed_sharded(fwd, bed, segment_center=2500, sink=TsvSink(out), timings=timings)
    assert fwd.reuse_sites == len(a) + len(b) + len(c)      # the dense chromosomes; chrS (1 site per ~80 bases) went per-window
    assert len(res["start"]) == len(rows) and {"bed_index", "bed_parse", "compute_enqueue", "sink"} <= set(timings)
    # the whole-file ingest (every rank parses the whole BED, the reference's one BedTool per process) gives the same rows and table
    whole_out = tmp_path / "pred_whole.tsv"
    whole = predict_bed_sharded(HipShardForward(model, fa, local_radius=r, local_order=3), bed, segment_center=2500, sink=TsvSink(whole_out),
                                ingest="whole")
    for key in ("chrom", "start", "end", "strand", "label", "prob", "order"):
        assert np.array_equal(whole[key], res[key]), key
    assert open(out, "rb").read() == open(whole_out, "rb").read()
    sub = rng.choice(len(rows), size=400, replace=False)
    for name, s in seqs.items():
        sel = sub[res["chrom"][sub] == name]
        want = _oracle_probs(orc, s, res["start"][sel], res["strand"][sel] == "-", r, R)
        assert np.abs(res["prob"][sel] - want).max() <= PROB_TOL, name
    # per-window driver on the same files: same rows, same order
    plain = predict_bed_sharded(HipShardForward(model, fa, local_radius=r, local_order=3, reuse=False), bed, segment_center=2500)
    assert np.array_equal(plain["start"], res["start"]) and np.array_equal(plain["chrom"], res["chrom"])
    assert np.abs(plain["prob"] - res["prob"]).max() <= PROB_TOL
    want_path = tmp_path / "want.tsv"
    write_predictions(res, want_path)
    assert open(out, "rb").read() == open(want_path, "rb").read()
    # a site whose base disagrees with its group is still caught (device check, verdict read one shard late)
    arr = np.frombuffer(seqs["chr10"].encode(), np.uint8)
    p_bad = int(np.nonzero(arr == ord("C"))[0][40])            # a C posing as a '+' site in the middle of chr10's A sites
    k_ins = next(i for i, (_, p, _) in enumerate(b) if p > p_bad)
    bad_rows = a + b[:k_ins] + [("chr10", p_bad, "+")] + b[k_ins:] + c + sparse
    bad_bed = tmp_path / "bad.bed"
    bad_bed.write_text("".join(f"{ch}\t{p}\t{p + 1}\t.\t0\t{st}\n" for ch, p, st in bad_rows))
    with pytest.raises(ValueError, match="different bases"):
        predict_bed_sharded(HipShardForward(model, fa, local_radius=r, local_order=3), bad_bed, segment_center=2500, collect=False)


def test_aligned_chromosomes_in_parts_write_the_gathered_route_s_table(tmp_path):
    """Round 6: a chromosome whose rows already are in the table's order is never gathered -- its rows go through in parts, each part
    checks its own (segment, strand) groups on the device, the border records are chained.  Real forward, parts of 700 rows, several
    segments per part and groups that run over part borders: the table is the gathered route's byte for byte, one rank's share of a
    3-rank run writes its third of it, and a wrong base in the middle of a group that spans parts fails the run."""
    from mural_amd import predict as P
    model, orc, r, R = _models(HUMAN[0])
    rng = np.random.default_rng(77)
    seqs = {"chr3": _genome(rng, 16_000), "chr7": _genome(rng, 5000, iupac=False)}
    fa = tmp_path / "g.fa"
    fa.write_text("".join(f">{k}\n" + "\n".join(s[i:i + 60] for i in range(0, len(s), 60)) + "\n" for k, s in seqs.items()))
    rows = []
    for name, s in seqs.items():
        arr = np.frombuffer(s.encode(), np.uint8)
        rows += [(name, int(p), "+" if arr[p] == ord("A") else "-") for p in range(len(arr)) if arr[p] in (ord("A"), ord("T"))]
    bed = tmp_path / "s.bed"
    bed.write_text("".join(f"{ch}\t{p}\t{p + 1}\t.\t{i % 4}\t{st}\n" for i, (ch, p, st) in enumerate(rows)))
    old = P._ALIGNED_BLOCKS, P._ALIGNED_PART_ROWS
    try:
        tables = {}
        for aligned in (True, False):
            P._ALIGNED_BLOCKS, P._ALIGNED_PART_ROWS = aligned, 700
            T = {}
            out = tmp_path / ("t%d.tsv" % aligned)
            n = P.predict_bed_sharded(P.HipShardForward(model, fa, local_radius=r, local_order=3), bed, segment_center=900, sink=P.TsvSink(out),
                                      collect=False, timings=T)
            assert n == len(rows) and T.get("aligned_shards", 0) == (2 if aligned else 0)
            tables[aligned] = open(out, "rb").read()
        assert tables[True] == tables[False] and tables[True].count(b"\n") == len(rows) + 1
        P._ALIGNED_BLOCKS = True
        # rank 1 of 3 (no process group): its part file holds its block's rows of both chromosomes, in the table's order
        share = tmp_path / "share.tsv"
        P.predict_bed_sharded(P.HipShardForward(model, fa, local_radius=r, local_order=3), bed, segment_center=900,
                              sink=P.TsvSink(share, parts=(1, 3)), collect=False, emulate=(1, 3))
        lines = tables[True].split(b"\n")[1:-1]
        by_chrom = {c: [ln for ln in lines if ln.split(b"\t")[0] == c.encode()] for c in seqs}
        want = []
        for c in sorted(seqs):
            lo, hi = P.shard_bounds(len(by_chrom[c]), 1, 3)
            want += by_chrom[c][lo:hi]
        assert open(str(share) + ".part0001", "rb").read() == b"".join(ln + b"\n" for ln in want)
        # a C posing as a '+' site in the middle of chr3: its group runs over several parts
        arr = np.frombuffer(seqs["chr3"].encode(), np.uint8)
        p_bad = int(np.nonzero(arr == ord("C"))[0][1500])
        bad_rows = sorted([rw for rw in rows if rw[0] == "chr3"] + [("chr3", p_bad, "+")], key=lambda rw: rw[1])
        bad = tmp_path / "bad.bed"
        bad.write_text("".join(f"{ch}\t{p}\t{p + 1}\t.\t0\t{st}\n" for ch, p, st in bad_rows))
        out = tmp_path / "bad.tsv"
        with pytest.raises(ValueError, match="different bases"):
            P.predict_bed_sharded(P.HipShardForward(model, fa, local_radius=r, local_order=3), bad, segment_center=900, sink=P.TsvSink(out),
                                  collect=False)
        assert not out.exists()
    finally:
        P._ALIGNED_BLOCKS, P._ALIGNED_PART_ROWS = old


def test_zero_row_rank_blocks_through_the_real_forward(tmp_path):
    """A rank whose block of a shard is empty (more ranks than rows) calls the real forward with zero sites: both entries return
    (0, n_class) / (0, n_class + 1) without touching the device queue in a way that breaks the next call."""
    from mural_amd.data import PackedGenome
    from mural_amd.predict import HipShardForward
    model, orc, r, R = _models(HUMAN[0])
    rng = np.random.default_rng(62)
    seq = _genome(rng, 6000, iupac=False)
    g = PackedGenome.from_sequence(seq, "cuda")
    z = torch.zeros(0, dtype=torch.int64, device="cuda")
    zs = torch.zeros(0, dtype=torch.uint8, device="cuda")
    with torch.no_grad():
        assert model.forward_packed(g, z, zs, local_radius=r, local_order=3).shape == (0, 4)
        assert model.forward_packed_reuse(g, z, zs, local_radius=r, local_order=3).shape == (0, 4)
    fa = tmp_path / "g.fa"
    fa.write_text(">c\n" + seq + "\n")
    fwd = HipShardForward(model, fa, local_radius=r, local_order=3)
    assert fwd("c", np.zeros(0, np.int64), np.zeros(0, np.uint8)).shape == (0, 5)
    pos = np.arange(1000, 1300)
    got = fwd("c", pos, np.zeros(300, np.uint8))[:, :4].cpu().numpy()
    assert np.abs(got - _oracle_probs(orc, seq, pos, np.zeros(300, bool), r, R)).max() <= PROB_TOL


# ------------------------------------------------------------------------------------------------------------------
# BASELINE config 1 on the reference's example material (examples/snv/examples.sh, Example 3): the rows of
# examples/snv/data/validation.sorted.bed below 400 kb on the seeded synthetic chr2L with planted focal bases (the example's
# data/seq.fa is not shipped), examples/snv/models/checkpoint_6 + its model.fdiri_cal.pkl.  G16 (config1_example.npz) holds what the
# reference's OWN pipeline wrote for them: prepare_dataset_np -> generate_data_batches -> model_predict_m -> softmax ->
# FullDirichletCalibrator.predict_proba -> sort_values -> to_csv('%.4g') (oracle/make_golden.py: g16_config1).
# ------------------------------------------------------------------------------------------------------------------
def _config1_files(tmp_path, fx, gz):
    import gzip
    rows = [ln.split("\t") for ln in str(fx["bed"]).split("\n") if ln]
    rng = np.random.default_rng(int(fx["genome_seed"]))
    seq = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, size=int(fx["genome_len"]))].copy()
    for c, s_, e_, name, score, strand in rows:
        seq[int(s_)] = ord("A") if strand == "+" else ord("T")
    fasta_text = (">chr2L synthetic\n" + "\n".join(seq.tobytes().decode()[i:i + 60] for i in range(0, len(seq), 60)) + "\n").encode()
    fa, bed = tmp_path / ("seq.fa.gz" if gz else "seq.fa"), tmp_path / ("validation.bed.gz" if gz else "validation.bed")
    fa.write_bytes(gzip.compress(fasta_text) if gz else fasta_text)
    bed.write_bytes(gzip.compress(str(fx["bed"]).encode()) if gz else str(fx["bed"]).encode())
    return fa, bed, len(rows)


def _table_fields(text):
    lines = text.rstrip("\n").split("\n")
    return lines[0], [ln.split("\t") for ln in lines[1:]]


@pytest.mark.parametrize("gz", [False, True])
def test_config1_example_files_to_calibrated_table(tmp_path, gz):
    from mural_amd.predict import HipShardForward, TsvSink, predict_bed_sharded
    fx = U.load("config1_example.npz")
    model, _, r, R = _models("snv_pretrained_example_ckpt6.npz")
    assert (r, R) == (int(fx["hp"][0]), int(fx["hp"][2]))
    fa, bed, n = _config1_files(tmp_path, fx, gz)
    w = fx["dirichlet_w"]
    # (a) softmax probabilities in bed_reader order against the reference's model_predict_m + softmax
    res = predict_bed_sharded(HipShardForward(model, fa, r, 3), bed, segment_center=int(fx["hp"][3]))
    assert len(res["start"]) == n
    assert np.abs(res["prob"] - fx["softmax"]).max() <= PROB_TOL
    # (b) files -> calibrated table, calibration on the host in the sink and on the device behind the head: every field but the
    # probabilities byte-identical to the reference's table, the probabilities within the tolerance -- a '%.4g' field may differ by a
    # unit of its last digit where a 1e-6 difference straddles a rounding boundary, and only there
    head_want, want = _table_fields(str(fx["table_calibrated"]))
    for mode in ("sink", "device"):
        out = tmp_path / f"pred_{mode}.tsv"
        if mode == "sink":
            predict_bed_sharded(HipShardForward(model, fa, r, 3), bed, segment_center=int(fx["hp"][3]), collect=False,
                                sink=TsvSink(out, dirichlet_weights=w))
        else:
            predict_bed_sharded(HipShardForward(model, fa, r, 3, dirichlet_weights=w), bed, segment_center=int(fx["hp"][3]), collect=False,
                                sink=TsvSink(out))
        head, got = _table_fields(open(out).read())
        assert head == head_want and len(got) == len(want) == n
        flips = 0
        for g, t in zip(got, want):
            assert g[:5] == t[:5]
            for a, b in zip(g[5:], t[5:]):
                if a != b:
                    flips += 1
                    unit = 10.0 ** (np.floor(np.log10(max(float(b), 1e-300))) - 3)      # one unit of the 4th significant digit
                    assert abs(float(a) - float(b)) <= 1.001 * unit, (g, t)
        assert flips <= n * 4 // 100, flips
    # (c) the reference's own softmax rows through this library's calibrator + sorter + '%.4g' writer: byte-identical tables
    from mural_amd.data.ingest import write_predictions
    fed = dict(res, prob=fx["softmax"])
    for name, kw in (("table_softmax", {}), ("table_calibrated", {"dirichlet_weights": w})):
        path = tmp_path / (name + ".tsv")
        write_predictions(fed, path, **kw)
        assert open(path).read() == str(fx[name]), name
