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
