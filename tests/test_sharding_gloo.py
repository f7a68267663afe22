"""Multi-process path on CPU: world_size 2, gloo.  Covers the shard split, the double-buffered
result gather to rank 0 and the max-over-ranks timing used by bench.py --gpus N."""
import importlib
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def test_shard_range_partitions():
    sharding = importlib.import_module("cart-pole-mpc_amd.sharding")
    for total, world in ((2097152, 8), (10, 3), (7, 8), (0, 2), (262144, 1)):
        spans = [sharding.shard_range(total, r, world) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == total
        for (a, b), (c, d) in zip(spans, spans[1:]):
            assert b == c and a <= b
        sizes = [b - a for a, b in spans]
        assert max(sizes) - min(sizes) <= 1
    assert [sharding.shard_range(2097152, r, 8) for r in (0, 7)] == [(0, 262144), (1835008, 2097152)]
    with pytest.raises(ValueError):
        sharding.shard_range(10, 2, 2)


def _worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sharding = importlib.import_module("cart-pole-mpc_amd.sharding")
    n_rows, total = 5, 64
    lo, hi = sharding.shard_range(total, rank, world)
    b_local = hi - lo
    g = sharding.ResultGather(n_rows, b_local, torch.float32, "cpu", dst=0, depth=2)
    ok = True
    for step in range(5):  # more steps than buffers: exercises the reuse of a slot
        # "results" of this rank's shard: value encodes (row, global problem index, step)
        block = (torch.arange(n_rows).reshape(-1, 1) * 1000 + torch.arange(lo, hi).reshape(1, -1)).float() + step * 0.5
        slot = g.submit(block)
        g.wait_slot(slot)
        if rank == 0:
            full = g.assembled(slot)
            want = (torch.arange(n_rows).reshape(-1, 1) * 1000 + torch.arange(total).reshape(1, -1)).float() + step * 0.5
            ok = ok and torch.equal(full, want)
    g.finish()
    t = sharding.max_over_ranks(1.0 + rank, "cpu")
    ok = ok and t == float(world)
    dist.barrier()
    with open(os.path.join(tmp, "ok%d" % rank), "w") as fh:
        fh.write("1" if ok else "0")
    dist.destroy_process_group()


def test_gather_world_size_2_gloo(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        assert (tmp_path / ("ok%d" % r)).read_text() == "1"


def test_single_process_passthrough():
    sharding = importlib.import_module("cart-pole-mpc_amd.sharding")
    g = sharding.ResultGather(3, 4, torch.float32, "cpu")
    g.submit(torch.zeros(3, 4))
    g.finish()
    assert g.assembled(0) is None
    assert sharding.max_over_ranks(2.5, "cpu") == 2.5
