"""The data-parallel training exchange (read_amd/ddp.py) at world_size 2 on gloo, CPU tensors: the flat gradient arena's
all-reduce equals the mean of the ranks' gradients, the ragged sparse descriptor pairs arrive in rank order scaled by 1/world,
and a full "train loop" of two ranks on half batches each follows the single-process full-batch trajectory
(reference semantics: nn.DataParallel + per-replica loss averaged, /root/reference/train.py:138-139)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from read_amd import ddp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _net(seed=0):
    torch.manual_seed(seed)
    return torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3, padding=1), torch.nn.BatchNorm2d(8), torch.nn.ELU(),
                               torch.nn.Conv2d(8, 5, 1, bias=False))


def _data(seed=1):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(4, 3, 6, 7, generator=g), torch.randn(4, 5, 6, 7, generator=g)


class _Tex:                                   # the two members of PointTexture that DataParallelStep touches
    def __init__(self, C):
        self.texture_ = torch.zeros(1, C, 50)
        self._pending = []

    def take_pending(self):
        if not self._pending:
            return None
        ids = torch.cat([p[0] for p in self._pending])
        g = torch.cat([p[1] for p in self._pending])
        self._pending = []
        return ids, g


def _pairs(rank, step, C=8):
    g = torch.Generator().manual_seed(100 * step + rank)
    n = [0, 7][rank] if step == 1 else 5 + 3 * rank + step         # ragged; rank 0 has NO pairs in step 1
    return torch.randint(0, 50, (n,), generator=g, dtype=torch.int32), torch.randn(n, C, generator=g)


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        net = _net(seed=rank)                                       # different initial weights: the constructor broadcasts rank 0's
        net.eval()
        tex = _Tex(8)
        step = ddp.DataParallelStep(net, [tex])
        assert step.arena.flat.numel() % 64 == 0
        opt = torch.optim.SGD(net.parameters(), lr=0.1)
        x, y = _data()
        xs, ys = x[2 * rank:2 * rank + 2], y[2 * rank:2 * rank + 2]   # this rank's half of the batch
        gathered = []
        for it in range(3):
            loss = torch.nn.functional.huber_loss(net(xs), ys)
            loss.backward()
            ids, rows = _pairs(rank, it)
            if ids.numel():
                tex._pending.append((ids, rows))
            step.reduce()
            assert all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(step.arena.params, step.arena.views))
            pend = tex.take_pending()
            gathered.append(None if pend is None else (pend[0].numpy().copy(), pend[1].numpy().copy()))
            opt.step()
            opt.zero_grad()                                          # set_to_none: the arena re-attaches in the next reduce()
        ddp.sync_buffers(net)
        q.put((rank, {k: v.numpy().copy() for k, v in net.state_dict().items()}, gathered))
    finally:
        dist.destroy_process_group()


def test_two_ranks_follow_the_full_batch_trajectory():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=180) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single process, full batch, mean of the two half-batch losses (= DataParallel's averaged per-replica loss)
    net = _net(seed=0)
    net.eval()
    opt = torch.optim.SGD(net.parameters(), lr=0.1)
    x, y = _data()
    for it in range(3):
        loss = 0.5 * (torch.nn.functional.huber_loss(net(x[:2]), y[:2]) + torch.nn.functional.huber_loss(net(x[2:]), y[2:]))
        loss.backward()
        opt.step()
        opt.zero_grad()
    ref = net.state_dict()
    for rank, sd, _ in got:
        for k, v in ref.items():
            assert np.allclose(sd[k], v.numpy(), rtol=1e-5, atol=1e-6), f"rank {rank}: {k} left the full-batch trajectory"
    for k in ref:                                                    # replicas bit-identical to each other
        assert np.array_equal(got[0][1][k], got[1][1][k]), k
    # descriptor pairs: both ranks hold the same list = rank 0's pairs then rank 1's, rows halved
    for it in range(3):
        a, b = got[0][2][it], got[1][2][it]
        want_ids = torch.cat([_pairs(r, it)[0] for r in range(2)]).numpy()
        want_rows = torch.cat([_pairs(r, it)[1] for r in range(2)]).numpy() * 0.5
        for g in (a, b):
            assert np.array_equal(g[0], want_ids) and g[0].dtype == np.int32
            assert np.array_equal(g[1], want_rows)


def test_single_process_is_the_identity():
    net = _net()
    tex = _Tex(8)
    step = ddp.DataParallelStep(net, [tex])
    x, y = _data()
    torch.nn.functional.huber_loss(net(x), y).backward()
    want = [p.grad.clone() for p in net.parameters()]
    ids, rows = _pairs(1, 0)
    tex._pending.append((ids, rows))
    step.reduce()
    for p, w in zip(net.parameters(), want):
        assert torch.equal(p.grad, w)
    pend = tex.take_pending()
    assert torch.equal(pend[0], ids) and torch.equal(pend[1], rows)
    # ids survive the bit-cast through the fp32 payload column for the whole int32 range used (< 2^31)
    big = torch.tensor([0, 1, 2 ** 24 + 1, 2 ** 30 + 12345, 2 ** 31 - 1], dtype=torch.int32)
    assert torch.equal(big.view(torch.float32).view(torch.int32), big)


def _worker_diverged(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        net = _net(seed=rank)                                       # different weights AND different BatchNorm statistics per rank
        net[1].running_mean.fill_(float(rank + 1))
        unused = torch.nn.Parameter(torch.full((3,), float(rank)))   # a parameter no loss ever reaches
        net.register_parameter("unused", unused)
        tex = _Tex(8)
        tex.texture_ = torch.full((1, 8, 50), float(10 + rank))      # per-rank descriptor tables (a 'rand' init with per-rank seeds)
        tex._rows = "stale"
        step = ddp.DataParallelStep(net, [tex])
        start = (tex.texture_.clone(), net[1].running_mean.clone(), unused.detach().clone(), tex._rows)
        net.train()
        x, y = _data()
        xs, ys = x[2 * rank:2 * rank + 2], y[2 * rank:2 * rank + 2]
        opt = torch.optim.Adam(net.parameters(), lr=0.1, weight_decay=0.1)
        for it in range(2):
            torch.nn.functional.huber_loss(net(xs), ys).backward()
            step.reduce()
            assert unused.grad is None, "a parameter without a gradient on every rank must stay without one"
            opt.step()
            opt.zero_grad()
        q.put((rank, [t.numpy().copy() if torch.is_tensor(t) else t for t in start], net[1].running_mean.numpy().copy(),
               net[1].running_var.numpy().copy(), unused.detach().numpy().copy()))
    finally:
        dist.destroy_process_group()


def test_replicas_that_start_apart_are_brought_together():
    """the constructor broadcasts rank 0's descriptor tables and buffers as well as the weights (replicas that
    start from per-rank random textures would otherwise diverge for good); in .train() mode reduce() keeps the BatchNorm
    statistics on rank 0's; a parameter without a gradient on every rank keeps grad = None (Adam / weight decay leave it alone)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_diverged, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=180) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, start, mean, var, unused in got:
        assert np.all(start[0] == 10.0), f"rank {rank}: descriptor table is not rank 0's"
        assert np.all(start[1] == 1.0), f"rank {rank}: BatchNorm buffer is not rank 0's"
        assert np.all(start[2] == 0.0) and start[3] is None          # the unused parameter was broadcast too; the row cache dropped
        assert np.all(unused == 0.0), "weight decay moved a parameter that has no gradient anywhere"
    assert np.array_equal(got[0][2], got[1][2]) and np.array_equal(got[0][3], got[1][3])     # running statistics: one trajectory
    assert not np.all(got[0][2] == 1.0)                              # ... and it did move
