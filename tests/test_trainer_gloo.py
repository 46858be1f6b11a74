"""Data-parallel training host logic on CPU with world_size-2 gloo processes: parameter broadcast, the single flat
gradient all-reduce (trainer.py:106 -> :115 in the reference loop), the all-reduced epoch loss and the reference
checkpoint format.  The model kernels need a GPU; here the gradients are set by hand, which is all the reducer sees."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, tmp, q):
    import sys
    sys.path.insert(0, ROOT)
    import tss_with_dprnn_amd as pkg
    from tss_with_dprnn_amd import trainer
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(100 + rank)                    # replicas start different: the broadcast must fix that
        model = pkg.DPRNNTasNet(64, n_repeats=1)
        opt = torch.optim.Adam(model.parameters(), lr=1e-3)
        tr = trainer.DistributedTrainer(model, opt, clip_norm=5, device=torch.device("cpu"),
                                        new_checkpoints_path=tmp if rank == 0 else None)
        w0 = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
        # hand-made gradients: rank r holds (r + 1) * index pattern; one parameter has no grad on rank 1
        params = list(model.parameters())
        for i, p in enumerate(params):
            p.grad = torch.full_like(p, float((rank + 1) * (i + 1)))
        if rank == 1:
            params[3].grad = None
        tr.reducer.average()
        got = [float(p.grad.reshape(-1)[0]) for p in params]
        mean = tr._mean_over_ranks(10.0 * (rank + 1), 2)             # totals 10, 20 over 2 steps each -> 7.5
        path = tr._save_checkpoint(best=True)
        q.put((rank, w0.double().sum().item(), got, tr.reducer.numel, mean, path))
    finally:
        dist.destroy_process_group()


def test_two_rank_gradient_average_and_checkpoint(tmp_path):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, str(tmp_path), q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, s0, g0, n0, m0, path0), (r1, s1, g1, n1, m1, path1) = res
    assert s0 == s1                                       # identical weights after the broadcast
    assert g0 == g1                                       # identical gradients after the all-reduce
    want = [1.5 * (i + 1) for i in range(len(g0))]
    want[3] = 0.5 * 4                                     # rank 1 contributed zeros for that parameter
    assert np.allclose(g0, want)
    assert n0 == n1 == 677633
    assert m0 == m1 == 7.5
    assert path1 is None and os.path.exists(path0)
    cpt = torch.load(path0, map_location="cpu")
    assert set(cpt) == {"epoch", "optimizer", "model"}    # trainer.py:294-306
    assert "separation.dprnn_blocks.0.intra_rnn.rnn.weight_ih_l0" in cpt["model"]


def test_single_process_reducer_is_a_no_op():
    import tss_with_dprnn_amd as pkg
    from tss_with_dprnn_amd import trainer
    model = pkg.DPRNNTasNet(64, n_repeats=1)
    for p in model.parameters():
        p.grad = torch.ones_like(p)
    red = trainer.FlatGradAllReducer(model.parameters())
    red.average()
    assert red.flat is None and all(float(p.grad.reshape(-1)[0]) == 1.0 for p in model.parameters())


def _worker_buffers(rank, world, port, q):
    import sys
    sys.path.insert(0, ROOT)
    from tss_with_dprnn_amd import dprnn_spe, trainer
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(7)
        model = dprnn_spe.DPRNNSpeTasNet(64, n_repeats=1)
        tr = trainer.DistributedTrainer(model, torch.optim.Adam(model.parameters(), lr=1e-3), device=torch.device("cpu"))
        bufs = [b for b in model.buffers() if b.dtype.is_floating_point]
        for b in bufs:                                  # per-rank BatchNorm running statistics drift apart in training
            b.data.fill_(float(rank + 1))
        tr._sync_buffers()                              # what eval() and _save_checkpoint() do first
        same = all(bool((b == 1.0).all()) for b in bufs)
        err = None
        try:                                            # loaders of different length would deadlock in the all-reduce
            tr._check_equal_steps([0] * (3 + rank))
        except RuntimeError as e:
            err = str(e)
        tr._check_equal_steps([0] * 3)                  # equal lengths pass
        q.put((rank, same, len(bufs), err))
    finally:
        dist.destroy_process_group()


def test_two_rank_buffer_sync_and_step_count_check():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_buffers, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, same, n, err in res:
        assert same and n >= 12                          # 3 ResBlocks x 2 BatchNorm1d x (running_mean, running_var)
        assert err is not None and "DistributedSampler" in err
