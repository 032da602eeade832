"""The prediction-table writer (csrc/tsv.hip, host side): byte-identical to the reference's
``pred_df.sort_values(['chrom', 'start']); pred_df.to_csv(sep='\\t', float_format='%.4g', index=False)``
(MuRaL/scripts/run_predict.py:236-238).  pandas is the checker here, never the product path."""
import ctypes as C
import io

import numpy as np
import pandas as pd
import pytest

from mural_amd import _lib
from mural_amd.predict import TsvSink, format_rows_host, write_predictions

from . import _util as U


def _g4(v):
    buf = C.create_string_buffer(16)
    n = _lib.lib().mural_tsv_format_g4(float(v), buf)
    return buf.raw[:n].decode()


def test_g4_known_answers():
    cases = [(-0.0, "-0")] + list({0.0: "0", 1.0: "1", 1e-5: "1e-05", 1e-4: "0.0001", 9999.5: "1e+04", 9999.4999: "9999", 99995.0: "1e+05",
             12345.0: "1.234e+04", 1234.5: "1234", 1235.5: "1236", 1e16: "1e+16", 5e-324: "4.941e-324", 0.99995: "1",
             0.00012345: "0.0001234", 1.7976931348623157e308: "1.798e+308", float("inf"): "inf", -float("inf"): "-inf",
             0.1: "0.1", 0.25: "0.25", 123.456: "123.5", -2.5e-7: "-2.5e-07"}.items())
    for v, want in cases:
        assert want == "%.4g" % v          # the table itself is what CPython prints
        assert _g4(v) == want, v
    assert _g4(float("nan")) == ""         # pandas' na_rep


def test_g4_matches_python_on_random_bit_patterns_and_near_ties():
    rng = np.random.default_rng(11)
    vals = [rng.integers(0, 2 ** 64, size=60000, dtype=np.uint64).view(np.float64),
            rng.integers(0, 2 ** 32, size=60000, dtype=np.uint32).view(np.float32).astype(np.float64),
            rng.random(60000), np.exp(rng.uniform(-60, 3, 60000))]
    near = []
    for e in range(-322, 306, 3):                      # decimal ties N.5 x 10^e and their floating-point neighbours
        for n4 in rng.integers(1000, 10000, size=6):
            try:
                v = float(f"{n4}5e{e - 1}")
            except OverflowError:
                continue
            for k in (-2, -1, 0, 1, 2):
                w = v
                for _ in range(abs(k)):
                    w = np.nextafter(w, np.inf if k > 0 else -np.inf)
                near.append(float(w))
    vals.append(np.array(near))
    bad = []
    for arr in vals:
        for v in arr:
            v = float(v)
            if v != v or v in (float("inf"), -float("inf")):
                continue
            if _g4(v) != "%.4g" % v:
                bad.append(v)
    assert not bad, bad[:5]


def _pandas_table(chrom, start, end, strand, label, prob):
    df = pd.concat((pd.DataFrame({"chrom": chrom, "start": start, "end": end, "strand": strand}),
                    pd.DataFrame({"mut_type": np.asarray(label).astype(np.int64)}),
                    pd.DataFrame(prob, columns=["prob%d" % i for i in range(prob.shape[1])])), axis=1)
    df.sort_values(["chrom", "start"], inplace=True)
    df.reset_index(drop=True, inplace=True)
    buf = io.StringIO()
    df.to_csv(buf, sep="\t", float_format="%.4g", index=False)
    return buf.getvalue().encode()


def _random_rows(rng, n, k=4, dtype=np.float32):
    prob = rng.random((n, k)).astype(dtype)
    if n >= 100:
        prob[rng.integers(0, n, 40), rng.integers(0, k, 40)] = np.nan
    m = min(n // 3, 1500)
    prob[:m] = np.exp(rng.uniform(-100, 10, (m, k))).astype(dtype)
    edge = np.array([1e-5, 1e-4, 9.9995e-5, 0.99995, 1e16, 9.9995e15, 1e-45, 1.1754942e-38, 0.0, -0.0, np.inf, -np.inf], dtype)
    e = min(300, n - m)
    prob[m:m + e] = rng.choice(edge, (e, k))
    chrom = rng.choice(np.array(["chr1", "chr10", "chr2", "01", "1", "X"], object), n)
    start = rng.integers(0, 50000, n)
    return {"chrom": chrom, "start": start, "end": start + rng.integers(1, 3, n), "strand": rng.choice(np.array(["+", "-"], object), n),
            "label": rng.integers(0, k, n).astype(np.float32), "prob": prob}


@pytest.mark.parametrize("dtype,k", [(np.float32, 4), (np.float64, 4), (np.float32, 2), (np.float64, 8)])
def test_write_predictions_byte_identical_to_pandas(tmp_path, dtype, k):
    """1e5 random rows incl. NaN, subnormals, the 1e-5 / 1e16 notation boundaries, +-0, +-inf, numeric-looking chromosome names
    ('01', '1', '10': string order) and ties in (chrom, start) (stable order)."""
    rng = np.random.default_rng(k + (dtype == np.float64))
    res = _random_rows(rng, 100_000, k, dtype)
    path = tmp_path / "t.tsv"
    assert write_predictions(res, path) == 100_000
    assert path.read_bytes() == _pandas_table(res["chrom"], res["start"], res["end"], res["strand"], res["label"], res["prob"])


def test_write_predictions_matches_the_table_the_reference_wrote(tmp_path):
    fx = U.load("output.npz")            # G11: table written by the reference's own to_csv
    res = {"chrom": fx["chrom"].astype(object), "start": fx["start"], "end": fx["start"] + 1, "strand": fx["strand"].astype(object),
           "label": fx["label"].astype(np.float32), "prob": fx["prob"]}
    write_predictions(res, tmp_path / "p.tsv")
    assert (tmp_path / "p.tsv").read_text() == str(fx["table"])


def test_format_rows_host_threads_and_perm():
    rng = np.random.default_rng(5)
    r = _random_rows(rng, 200_000)
    st = (r["strand"] == "-").astype(np.uint8)
    perm = rng.permutation(200_000)
    one = format_rows_host(["chrZ"], None, r["start"], r["end"], st, r["label"], r["prob"], perm, threads=1)
    many = format_rows_host(["chrZ"], None, r["start"], r["end"], st, r["label"], r["prob"], perm, threads=7)
    assert one == many and one.count(b"\n") == 200_000
    first = one.split(b"\n", 1)[0].split(b"\t")
    i = perm[0]
    assert first[:5] == [b"chrZ", str(r["start"][i]).encode(), str(r["end"][i]).encode(), b"-" if st[i] else b"+",
                         str(int(r["label"][i])).encode()]
    assert format_rows_host(["c"], None, np.zeros(0, np.int64), np.zeros(0, np.int64), np.zeros(0, np.uint8), np.zeros(0, np.float32),
                            np.zeros((0, 4), np.float32)) == b""


def _shards(res, names_in_order):
    for nm in names_in_order:
        sel = res["chrom"] == nm
        yield {"chrom": res["chrom"][sel], "start": res["start"][sel], "end": res["end"][sel], "strand": res["strand"][sel],
               "label": res["label"][sel], "prob": res["prob"][sel]}


@pytest.mark.parametrize("arrival", [["01", "1", "X", "chr1", "chr10", "chr2"],          # ascending: streamed
                                     ["chr1", "chr2", "chr10", "X", "1", "01"],          # natural / arbitrary order: spooled + merged
                                     ["chr2", "chr1", "chr1b"]])
def test_tsv_sink_any_arrival_order_equals_pandas(tmp_path, arrival):
    rng = np.random.default_rng(8)
    res = _random_rows(rng, 5000)
    if "chr1b" in arrival:               # a chromosome delivered in two pieces, after a larger name was streamed
        res["chrom"] = rng.choice(np.array(["chr1", "chr2"], object), 5000)
        pieces = []
        sel1 = np.nonzero(res["chrom"] == "chr1")[0]
        for nm, idx in (("chr2", np.nonzero(res["chrom"] == "chr2")[0]), ("chr1", sel1[: len(sel1) // 2]), ("chr1", sel1[len(sel1) // 2:])):
            pieces.append({key: res[key][idx] for key in res})
        shards = pieces
    else:
        shards = list(_shards(res, arrival))
    sink = TsvSink(tmp_path / "s.tsv")
    for sh in shards:
        sink(sh)
    sink.close()
    want = _pandas_table(res["chrom"], res["start"], res["end"], res["strand"], res["label"], res["prob"])
    got = (tmp_path / "s.tsv").read_bytes()
    if "chr1b" in arrival:
        # two pieces of one chromosome: ties in start keep arrival order, which here is not the concatenated row order --
        # compare as sorted row sets per (chrom, start) instead
        assert sorted(got.split(b"\n")) == sorted(want.split(b"\n"))
        starts = [int(ln.split(b"\t")[1]) for ln in got.split(b"\n")[1:-1] if ln.startswith(b"chr1\t")]
        assert starts == sorted(starts)
    else:
        assert got == want


def test_tsv_sink_calibration_on_host_rows(tmp_path):
    from mural_amd.calibration import dirichlet_calibrate
    from mural_amd.data.ingest import poisson_calibrate
    rng = np.random.default_rng(9)
    p = rng.random((500, 4)).astype(np.float32)
    p /= p.sum(1, keepdims=True)
    w = rng.normal(size=(4, 5)) * 0.2 + np.hstack([np.eye(4), np.zeros((4, 1))])
    start = np.sort(rng.choice(10000, 500, replace=False))
    sh = {"chrom": np.array(["c"] * 500, object), "start": start, "end": start + 1, "strand": np.array(["+"] * 500, object),
          "label": np.zeros(500, np.float32), "prob": p}
    sink = TsvSink(tmp_path / "c.tsv", poisson=True, dirichlet_weights=w)
    sink(sh)
    sink.close()
    want = _pandas_table(sh["chrom"], start, start + 1, sh["strand"], sh["label"], poisson_calibrate(dirichlet_calibrate(p, w)))
    assert (tmp_path / "c.tsv").read_bytes() == want


def test_rows_calibrated_on_the_device_are_not_calibrated_again(tmp_path):
    """HipShardForward(model_type='indel') now applies the Poisson calibration by default and marks its shards
    `calibrated`; a sink or write_predictions that is asked to calibrate such rows refuses (it used to calibrate twice, silently)."""
    rng = np.random.default_rng(3)
    r = _random_rows(rng, 50)
    res = dict(r, calibrated=True)
    with pytest.raises(ValueError, match="calibrated already"):
        write_predictions(res, tmp_path / "a.tsv", poisson=True)
    with pytest.raises(ValueError, match="calibrated already"):
        write_predictions(res, tmp_path / "a.tsv", dirichlet_weights=np.hstack([np.eye(4), np.zeros((4, 1))]))
    assert write_predictions(res, tmp_path / "a.tsv") == 50                      # nothing asked: written as they are
    assert write_predictions(dict(r), tmp_path / "b.tsv", poisson=True) == 50    # uncalibrated rows: calibrated here, as before
    sink = TsvSink(tmp_path / "c.tsv", poisson=True)
    sel = r["chrom"] == "chr1"
    shard = {"chrom": "chr1", "start": r["start"][sel], "end": r["end"][sel], "strand": (r["strand"][sel] == "-").astype(np.uint8),
             "label": r["label"][sel], "prob": r["prob"][sel], "n_class": 4, "calibrated": True}
    with pytest.raises(ValueError, match="calibrated already"):
        sink(shard)
    sink.abort()


def test_config1_reference_rows_through_calibrator_and_writer_are_byte_identical(tmp_path):
    """G16 (BASELINE config 1 on the reference's example material, written by the reference's own run_predict pipeline): its softmax rows
    (model_predict_m order = bed_reader order) through this library's BED reader / row order, Dirichlet map, stable (chrom, start) sort
    and '%.4g' writer give the reference's two tables byte for byte."""
    import os
    from mural_amd.calibration import load_dirichlet_weights
    from mural_amd.data import ingest as I
    from tests import _util as U
    fx = U.load("config1_example.npz")
    bed = tmp_path / "validation.bed"
    bed.write_text(str(fx["bed"]))
    sites = I.read_bed(bed)
    order, _ = I.bed_order(sites, int(fx["hp"][3]))
    res = {"chrom": np.asarray(sites.chrom_names, object)[sites.chrom_id[order]], "start": sites.start[order], "end": sites.end[order],
           "strand": np.where(sites.strand[order] == 1, "-", "+").astype(object), "label": sites.score[order], "prob": fx["softmax"]}
    w = fx["dirichlet_w"]
    for name, kw in (("table_softmax", {}), ("table_calibrated", {"dirichlet_weights": w})):
        path = tmp_path / (name + ".tsv")
        assert write_predictions(res, path, **kw) == len(order)
        assert open(path).read() == str(fx[name]), name
    pkl = "/root/reference/examples/snv/models/checkpoint_6/model.fdiri_cal.pkl"
    if os.path.exists(pkl):      # (build container only: the shipped calibrator through the jax-free reader)
        assert np.array_equal(load_dirichlet_weights(pkl), w)
