"""The device side of the prediction-table writer (csrc/tsv.hip): the row-format kernel against the host formatter and against the
reference's own writer (pandas sort_values + to_csv('%.4g'), MuRaL/scripts/run_predict.py:236-238), the focal-base check kernel,
and TsvSink fed with device shards."""
import ctypes as C

import numpy as np
import pytest
import torch

from mural_amd import _lib
from mural_amd.predict import TsvSink, _name_table, _tsv_struct, check_focal_groups, format_rows_host
from tests.test_tsv import _pandas_table, _random_rows

pytestmark = pytest.mark.gpu


def _format_device(name, start, end, strand, label, prob, k, perm=None):
    lib = _lib.lib()
    dev = prob.device
    n = start.shape[0]
    names = _name_table([name])
    t = _tsv_struct(names, 1, None, start.data_ptr(), end.data_ptr(), strand.data_ptr(), label.data_ptr(), prob.data_ptr(),
                    prob.dtype == torch.float64, k, prob.stride(0), None if perm is None else perm.data_ptr(), n)
    bound = int(lib.mural_tsv_row_bound(C.byref(t)))
    text = torch.empty(max(n * bound, 1), dtype=torch.uint8, device=dev)
    count = torch.zeros(1, dtype=torch.int64, device=dev)
    ws = torch.empty(int(lib.mural_tsv_format_workspace_bytes(n)) + 256, dtype=torch.uint8, device=dev)
    _lib.check(lib.mural_tsv_format_device(C.byref(t), text.data_ptr(), text.numel(), count.data_ptr(), ws.data_ptr(), ws.numel(),
                                          _lib.current_stream_ptr(dev)))
    return text[:int(count.item())].cpu().numpy().tobytes()


@pytest.mark.parametrize("dtype,k,n", [(np.float32, 4, 100_000), (np.float64, 4, 100_000), (np.float32, 2, 1000), (np.float64, 8, 5000),
                                       (np.float32, 4, 1), (np.float32, 4, 257), (np.float32, 40, 3000)])
def test_device_formatter_equals_host_formatter_and_pandas(dtype, k, n):
    """Random rows incl. NaN, float32 / float64 subnormals, the 1e-5 / 1e16 notation boundaries, +-0, +-inf; the probabilities sit in
    a (n, k + 1) matrix like the gathered shard (row stride k + 1), rows are emitted through a permutation."""
    rng = np.random.default_rng(n + k)
    r = _random_rows(rng, n, k, dtype)
    st = (r["strand"] == "-").astype(np.uint8)
    wide = np.concatenate([r["prob"], np.full((n, 1), 7, dtype)], axis=1)
    perm = np.argsort(r["start"], kind="stable")
    want = format_rows_host(["chr12"], None, r["start"], r["end"], st, r["label"], r["prob"], perm)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()      # noqa: E731
    got = _format_device("chr12", dev(r["start"]), dev(r["end"]), dev(st), dev(r["label"]), dev(wide), k, dev(perm))
    assert got == want
    if k <= 8:
        chrom = np.array(["chr12"] * n, object)
        table = _pandas_table(chrom, r["start"], r["end"], r["strand"], r["label"], r["prob"])
        assert table.split(b"\n", 1)[1] == got


def test_device_formatter_empty_and_long_names():
    z = lambda dt: torch.zeros(0, dtype=dt, device="cuda")      # noqa: E731
    assert _format_device("c", z(torch.int64), z(torch.int64), z(torch.uint8), z(torch.float32), torch.zeros((0, 5), device="cuda"), 4) == b""
    name = "scaffold_" + "x" * 200
    one = lambda v, dt: torch.tensor([v], dtype=dt, device="cuda")      # noqa: E731
    got = _format_device(name, one(2 ** 62, torch.int64), one(2 ** 62 + 1, torch.int64), one(1, torch.uint8), one(3, torch.float32),
                         torch.tensor([[0.25, 1e-7]], device="cuda"), 2)
    assert got == f"{name}\t{2 ** 62}\t{2 ** 62 + 1}\t-\t3\t0.25\t1e-07\n".encode()


def test_focal_group_check_kernel_matches_host_check():
    rng = np.random.default_rng(4)
    n = 300_000
    group = np.sort(rng.integers(0, 4000, n)).astype(np.int64)
    focal = (group % 4).astype(np.float32)
    rows = np.concatenate([rng.random((n, 4)).astype(np.float32), focal[:, None]], axis=1)
    lib = _lib.lib()

    def run(rows_np, f64=False):
        t = torch.from_numpy(rows_np.astype(np.float64 if f64 else np.float32)).cuda()
        g = torch.from_numpy(group).cuda()
        status = torch.zeros(1, dtype=torch.int32, device="cuda")
        _lib.check(lib.mural_focal_group_check(t.data_ptr(), int(f64), t.stride(0), 4, g.data_ptr(), n, status.data_ptr(),
                                              _lib.current_stream_ptr(t.device)))
        return int(status.item())

    assert run(rows) == 0 and run(rows, True) == 0
    check_focal_groups(rows[:, 4].astype(np.int64), group)
    bad = rows.copy()
    i = int(np.nonzero(group[1:] == group[:-1])[0][1000]) + 1       # second row of some group
    bad[i, 4] = (bad[i, 4] + 1) % 4
    assert run(bad) == 1 and run(bad, True) == 1
    with pytest.raises(ValueError, match="different bases"):
        check_focal_groups(bad[:, 4].astype(np.int64), group)


@pytest.mark.parametrize("calibrated", [False, True])
def test_tsv_sink_with_device_shards_equals_pandas(tmp_path, calibrated, monkeypatch):
    """Device shards (the gathered (n, k + 1) matrix of the sharded driver) through TsvSink: device sort, device calibration,
    several pieces per shard, writer thread -- byte-identical to the pandas table of the same rows."""
    from mural_amd.calibration import dirichlet_calibrate
    from mural_amd.data.ingest import poisson_calibrate
    monkeypatch.setattr(TsvSink, "PIECE_ROWS", 4096)             # several pieces and buffer recycling on a small input
    rng = np.random.default_rng(21)
    n = 50_000
    r = _random_rows(rng, n)
    prob = rng.random((n, 4)).astype(np.float32)
    prob /= prob.sum(1, keepdims=True)
    r["prob"] = prob
    w = rng.normal(size=(4, 5)) * 0.2 + np.hstack([np.eye(4), np.zeros((4, 1))]) if calibrated else None
    sink = TsvSink(tmp_path / "d.tsv", poisson=calibrated, dirichlet_weights=w)
    for nm in sorted(set(r["chrom"].tolist())):
        sel = r["chrom"] == nm
        rows = torch.from_numpy(np.concatenate([prob[sel], np.zeros((sel.sum(), 1), np.float32)], axis=1)).cuda()
        sink({"chrom": nm, "start": torch.from_numpy(r["start"][sel]).cuda(), "end": torch.from_numpy(r["end"][sel]).cuda(),
              "strand": torch.from_numpy((r["strand"][sel] == "-").astype(np.uint8)).cuda(),
              "label": torch.from_numpy(r["label"][sel]).cuda(), "prob": rows, "n_class": 4})
    sink.close()
    want_prob = prob
    if calibrated:
        want_prob = poisson_calibrate(dirichlet_calibrate(prob, w))
    want = _pandas_table(r["chrom"], r["start"], r["end"], r["strand"], r["label"], want_prob)
    got = (tmp_path / "d.tsv").read_bytes()
    if not calibrated:
        assert got == want
    else:        # device calibration differs from numpy in the last bits of float32 log: a rare '%.4g' rounding flip at most
        a, b = got.split(b"\n"), want.split(b"\n")
        assert len(a) == len(b) and sum(x != y for x, y in zip(a, b)) <= n // 200
    assert sink.writer_seconds()["bytes"] == len(got) - len(got.split(b"\n", 1)[0]) - 1


@pytest.mark.parametrize("long_name", [False, True])
@pytest.mark.parametrize("world", [2, 8])
def test_part_file_slices_of_device_shards_concatenate_to_the_single_writer_table(tmp_path, world, long_name, monkeypatch):
    """The part-file mode on the device path: rank r of `world` sorts each shard and formats only its slice of the sorted rows
    (TsvSink(parts=(r, world)) without a process group writes just that rank's part file, with its per-shard byte counts).  The
    slices of all ranks, strung together shard by shard in rank order behind the header -- what rank 0 does at close() under
    torch.distributed (tests/test_dist_gloo.py runs that on the host path) -- are byte-identical to the single-writer table."""
    monkeypatch.setattr(TsvSink, "PIECE_ROWS", 2048)
    rng = np.random.default_rng(33)
    n = 30_000
    r = _random_rows(rng, n)
    prob = rng.random((n, 4)).astype(np.float32)
    prob /= prob.sum(1, keepdims=True)
    if long_name:      # a later chromosome name > 24 bytes longer than the staging allows replaces the writer thread
        r["chrom"] = np.array([c if c != "chr2" else "chr2_" + "scaffold" * 8 for c in r["chrom"]], object)      # mid-run; the
    names = sorted(set(r["chrom"].tolist()))                                          # earlier shards' byte counts must survive it

    def feed(sink):
        writers = []                   # (references kept: ids stay unique)
        for nm in names:
            writers.append(sink._writer)
            sel = r["chrom"] == nm
            rows = torch.from_numpy(np.concatenate([prob[sel], np.zeros((sel.sum(), 1), np.float32)], axis=1)).cuda()
            sink({"chrom": nm, "start": torch.from_numpy(r["start"][sel]).cuda(), "end": torch.from_numpy(r["end"][sel]).cuda(),
                  "strand": torch.from_numpy((r["strand"][sel] == "-").astype(np.uint8)).cuda(),
                  "label": torch.from_numpy(r["label"][sel]).cuda(), "prob": rows, "n_class": 4})
        writers.append(sink._writer)
        sink.close()
        return len({id(w) for w in writers if w is not None})

    single = TsvSink(tmp_path / "single.tsv")
    assert feed(single) == (2 if long_name else 1)
    want = (tmp_path / "single.tsv").read_bytes()
    parts, counts = [], []
    for rank in range(world):
        sink = TsvSink(tmp_path / "p.tsv", parts=(rank, world))
        feed(sink)
        parts.append((tmp_path / ("p.tsv.part%04d" % rank)).read_bytes())
        per_shard = dict(getattr(sink, "_writer_shard_bytes", {}))
        counts.append([per_shard.get(i, 0) for i in range(len(names))])
        assert sum(counts[-1]) == len(parts[-1])
    got, offs = want.split(b"\n", 1)[0] + b"\n", [0] * world
    for i in range(len(names)):
        for rank in range(world):
            got += parts[rank][offs[rank]:offs[rank] + counts[rank][i]]
            offs[rank] += counts[rank][i]
    assert got == want
