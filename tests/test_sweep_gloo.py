"""CPU, world_size 2, gloo: the pose-sharded sweep (read_amd/sweep.py) — sharding, scene broadcast and
the frame all-gather.  The per-frame renderer is replaced by the oracle rasteriser here (tests may
use the oracle); on the GPU the same driver wraps FrameRenderer."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle
from read_amd import camera, synthetic
from read_amd import sweep
from read_amd.sweep import broadcast_scene, render_sweep, shard_indices

W, H, N, POSES = 48, 32, 4000, 7


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _frame(xyz, proj, k):
    M = camera.total_matrix(proj, synthetic.sweep_pose(k))[0]
    idx, dep = oracle.raster_level(xyz, M, W, H)
    return torch.from_numpy(np.stack([idx.astype(np.float32), dep], -1))


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        xyz = torch.from_numpy(synthetic.make_cloud(N)) if rank == 0 else torch.zeros(N, 3)
        broadcast_scene([xyz], src=0)
        proj = synthetic.make_proj(W, H, f=30.0)
        mine = shard_indices(POSES, rank, world)
        frames = render_sweep(lambda k: _frame(xyz.numpy(), proj, k), POSES, (H, W, 2), torch.device("cpu"))
        q.put((rank, mine, frames.numpy()))
    finally:
        dist.destroy_process_group()


def test_two_rank_sweep_matches_single_process():
    assert shard_indices(7, 0, 2) == [0, 2, 4, 6] and shard_indices(7, 1, 2) == [1, 3, 5]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    xyz, proj = synthetic.make_cloud(N), synthetic.make_proj(W, H, f=30.0)
    ref = np.stack([_frame(xyz, proj, k).numpy() for k in range(POSES)])
    for rank, mine, frames in got:
        assert frames.shape == (POSES, H, W, 2)
        assert np.array_equal(frames, ref), f"rank {rank} does not hold the full sweep in pose order"


def test_single_process_sweep():
    xyz, proj = synthetic.make_cloud(N), synthetic.make_proj(W, H, f=30.0)
    frames = render_sweep(lambda k: _frame(xyz, proj, k), 3, (H, W, 2), torch.device("cpu"))
    assert frames.shape == (3, H, W, 2)


# ---- the loop bench.py times (sweep.run_steps + FrameExchange + broadcast_scene_from_rank0), with a stub renderer ----
def _bench_worker(rank, world, port, q, mode):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cpu")
        made = []

        def make():                                            # rank 0 only: the scene every rank receives
            made.append(rank)
            return [torch.from_numpy(synthetic.make_cloud(N)), torch.arange(7, dtype=torch.uint8)]
        xyz, blob = sweep.broadcast_scene_from_rank0(make, dev)
        assert made == ([0] if rank == 0 else []) and torch.equal(blob, torch.arange(7, dtype=torch.uint8))
        proj = synthetic.make_proj(W, H, f=30.0)
        log = []

        def render_into(k, out):
            log.append(k)
            out.copy_(_frame(xyz.numpy(), proj, k))
        ex = sweep.FrameExchange((H, W, 2), dev, torch.float32, mode)
        warm, steps = 1, 4
        sweep.run_steps(render_into, ex, 0, warm, POSES)
        ex.drain()
        dist.barrier()
        seen = {}
        for i in range(warm, warm + steps):                    # same call as bench.py's timed region, one step at a time
            sweep.run_steps(render_into, ex, i, 1, POSES)
            fr = ex.frames(i)
            if mode == 'all' or rank == 0:
                seen[i] = fr.clone().numpy()
        ex.drain()
        flags = sweep.gather_objects(rank == 0 or None)            # bench.py's verified_ranks: one verdict per rank
        assert flags == [True, None]
        q.put((rank, log, seen))
    finally:
        dist.destroy_process_group()


def _run_bench_loop(mode):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bench_worker, args=(r, 2, port, q, mode)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    xyz, proj = synthetic.make_cloud(N), synthetic.make_proj(W, H, f=30.0)
    for rank, log, seen in got:
        assert log == [(i * 2 + rank) % POSES for i in range(5)]          # pose (i * world + rank) % n_poses
        if mode == 'root' and rank != 0:
            assert seen == {}
            continue
        for i, frames in seen.items():
            assert frames.shape == (2, H, W, 2)
            for r in range(2):
                assert np.array_equal(frames[r], _frame(xyz, proj, (i * 2 + r) % POSES).numpy()), (mode, rank, i, r)


def test_bench_loop_two_ranks_all_gather():
    _run_bench_loop('all')


def test_bench_loop_two_ranks_gather_to_root():
    _run_bench_loop('root')


def test_bench_loop_single_process_has_no_exchange():
    ex = sweep.FrameExchange((2, 2), torch.device("cpu"), torch.float32, 'all')
    assert ex.mode is None and ex.world == 1
    sweep.run_steps(lambda k, out: out.fill_(k), ex, 0, 3, POSES)
    assert float(ex.frames(2)[0, 0, 0]) == 2.0
    assert sweep.gather_objects(True) == [True]


# ---- a whole sweep whose length is not a multiple of the world size, through run_steps ----
def _ragged_worker(rank, world, port, q, n_poses, layout='interleave'):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cpu")
        xyz, = sweep.broadcast_scene_from_rank0(lambda: [torch.from_numpy(synthetic.make_cloud(N))], dev)
        proj = synthetic.make_proj(W, H, f=30.0)
        log = []

        def render_into(k, out):
            log.append(k)
            out.copy_(_frame(xyz.numpy(), proj, k))
        ex = sweep.FrameExchange((H, W, 2), dev, torch.float32, 'all')
        steps = sweep.sweep_steps(n_poses, world)
        stack = torch.zeros((n_poses, H, W, 2))
        filled = []
        for i in range(steps):
            sweep.run_steps(render_into, ex, i, 1, n_poses, layout=layout)
            fr = ex.frames(i)
            for r in range(world):
                k = sweep.pose_of_step(i, r, world, n_poses, layout)
                if not sweep.is_wrapped(i, r, world, n_poses, layout):   # the wrapped tail is a repeat of the head: dropped
                    stack[k].copy_(fr[r])
                    filled.append(k)
                else:
                    assert torch.equal(fr[r], stack[k])
        ex.drain()
        q.put((rank, log, filled, stack.numpy()))
    finally:
        dist.destroy_process_group()


def test_ragged_sweep_through_run_steps_two_ranks():
    n_poses = 5                                                  # 5 % 2 != 0: rank 1 wraps to pose 0 in the last step
    assert sweep.sweep_steps(n_poses, 2) == 3 and sweep.pose_of_step(2, 1, 2, n_poses) == 0
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ragged_worker, args=(r, 2, port, q, n_poses)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    xyz, proj = synthetic.make_cloud(N), synthetic.make_proj(W, H, f=30.0)
    ref = np.stack([_frame(xyz, proj, k).numpy() for k in range(n_poses)])
    for rank, log, filled, stack in got:
        assert log == ([0, 2, 4] if rank == 0 else [1, 3, 0])        # every step renders on every rank; the tail wraps
        assert filled == list(range(n_poses))                        # each pose of the sweep received exactly once, in order
        assert np.array_equal(stack, ref), f"rank {rank}: ragged sweep differs from the single-process frames"


def test_block_layout_keeps_every_rank_on_consecutive_poses():
    """'block' layout (bench.py --pose-layout block): rank r walks the contiguous block r * ceil(n / world) ..., so the
    rasteriser's warm start sees consecutive poses on every rank; every pose of a sweep_steps-long sweep is covered exactly
    once by the non-wrapped (step, rank) pairs — for ragged lengths and for both layouts."""
    for layout in sweep.LAYOUTS:
        for n_poses, world in ((256, 8), (5, 2), (7, 3), (10, 4), (3, 8), (256, 1)):
            steps = sweep.sweep_steps(n_poses, world)
            seen = [sweep.pose_of_step(i, r, world, n_poses, layout) for i in range(steps) for r in range(world)
                    if not sweep.is_wrapped(i, r, world, n_poses, layout)]
            assert sorted(seen) == list(range(n_poses)), (layout, n_poses, world)
            for r in range(world):
                walk = [sweep.pose_of_step(i, r, world, n_poses, layout) for i in range(steps)]
                stride = 1 if layout == 'block' else world
                assert all((b - a) % n_poses == stride % n_poses for a, b in zip(walk, walk[1:])), (layout, walk)
    assert [sweep.pose_of_step(i, 3, 8, 256, 'block') for i in range(3)] == [96, 97, 98]
    assert [sweep.pose_of_step(i, 3, 8, 256, 'interleave') for i in range(3)] == [3, 11, 19]
    with pytest.raises(ValueError):
        sweep.pose_of_step(0, 0, 2, 4, 'tiles')


def test_ragged_sweep_block_layout_two_ranks():
    n_poses = 5                                                  # blocks of 3: rank 0 poses 0 1 2, rank 1 poses 3 4 then wraps to 0
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ragged_worker, args=(r, 2, port, q, n_poses, 'block')) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    xyz, proj = synthetic.make_cloud(N), synthetic.make_proj(W, H, f=30.0)
    ref = np.stack([_frame(xyz, proj, k).numpy() for k in range(n_poses)])
    for rank, log, filled, stack in got:
        assert log == ([0, 1, 2] if rank == 0 else [3, 4, 0])
        assert sorted(filled) == list(range(n_poses)) and len(filled) == n_poses
        assert np.array_equal(stack, ref), f"rank {rank}: block-layout sweep differs from the single-process frames"


def test_run_steps_announces_the_next_pose():
    """run_steps(announce_next=True): the renderer is told the pose this rank renders NEXT (the rasteriser prepares that frame inside
    the current one's last launch) — for both layouts, a shard override (bench.py's single-GPU proxy of rank 3 of 8) and the wrap."""
    ex = sweep.FrameExchange((2, 2), torch.device("cpu"), torch.float32, None)
    for layout in sweep.LAYOUTS:
        seen = []
        sweep.run_steps(lambda k, out, nxt: seen.append((k, nxt)), ex, 0, 34, 256, layout, (3, 8), True)
        want = [sweep.pose_of_step(i, 3, 8, 256, layout) for i in range(35)]
        assert seen == list(zip(want[:-1], want[1:])), layout
    assert seen[0] == (96, 97) and seen[31] == (127, 128)          # block layout: rank 3 walks 96, 97, ... (and on past its block)
    plain = []
    sweep.run_steps(lambda k, out: plain.append(k), ex, 5, 3, 256)
    assert plain == [5, 6, 7]
