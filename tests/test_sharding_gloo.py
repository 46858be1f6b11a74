"""N > 1 host logic on CPU: world_size-2 gloo processes exercise the utterance sharding, the metric gather and the
max-over-ranks timing reduction that bench.py / a multi-GPU inference run use (no data-path collective exists)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_total, q):
    import sys
    sys.path.insert(0, ROOT)
    from tss_with_dprnn_amd import sharding
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        a, b = sharding.shard_range(n_total, rank, world)
        local = torch.arange(a, b, dtype=torch.float64) * 2.0          # "metric" of utterance i is 2*i
        full = sharding.gather_metric(local, n_total)
        slow = sharding.max_over_ranks(1.0 + rank)
        q.put((rank, (a, b), full.tolist(), slow))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [7, 64])
def test_two_rank_gather_and_max(n_total):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_total, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    ranges = [r[1] for r in res]
    assert ranges[0][0] == 0 and ranges[0][1] == ranges[1][0] and ranges[1][1] == n_total
    for _, _, full, slow in res:
        assert full == [2.0 * i for i in range(n_total)]
        assert slow == 2.0


def test_shard_range_is_a_partition():
    from tss_with_dprnn_amd.sharding import shard_range
    for n in (0, 1, 5, 64, 3000):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_shard_by_length_balances():
    from tss_with_dprnn_amd.sharding import shard_by_length
    import numpy as np
    lengths = np.random.default_rng(0).integers(24000, 111920, size=300).tolist()   # SURVEY.md 8d distribution
    parts = shard_by_length(lengths, 8)
    assert sorted(i for p in parts for i in p) == list(range(300))
    loads = [sum(lengths[i] for i in p) for p in parts]
    assert (max(loads) - min(loads)) / max(loads) < 0.02


def _inferencer_worker(rank, world, port, q):
    import sys
    sys.path.insert(0, ROOT)
    from tss_with_dprnn_amd.inferencer import BatchedInferencer
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        inf = BatchedInferencer(torch.nn.Identity(), "cpu")
        lengths = [24000, 112000, 40000, 24000, 64000, 88000, 31000]
        mine = inf._my_items(lengths)
        rows = torch.tensor([[float(i), 10.0 * i] for i in mine], dtype=torch.float64).reshape(-1, 2)
        table = inf._gather(mine, rows, len(lengths))
        final = inf._finish(table, 0.0)
        q.put((rank, mine, table.tolist(), final))
    finally:
        dist.destroy_process_group()


def test_two_rank_inferencer_partition_and_gather():
    """BatchedInferencer's host logic with two ranks: every item evaluated exactly once, length-balanced, and the
    per-utterance table / final means identical on both ranks (inferencer.py:84-104 aggregate)."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_inferencer_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, m0, t0, f0), (_, m1, t1, f1) = res
    assert sorted(m0 + m1) == list(range(7)) and m0 and m1
    assert t0 == t1 == [[float(i), 10.0 * i] for i in range(7)]
    assert f0 == f1 and abs(f0["si_sdr"] - 3.0) < 1e-12 and abs(f0["si_sdr_imp"] - (3.0 - 30.0)) < 1e-12
