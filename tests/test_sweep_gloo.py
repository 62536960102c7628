"""CPU, world_size 2, gloo: the pose-sharded sweep (read_amd/sweep.py) — sharding, scene broadcast and
the frame all-gather.  The per-frame renderer is replaced by the oracle rasteriser here (tests may
use the oracle); on the GPU the same driver wraps FrameRenderer."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle
from read_amd import camera, synthetic
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
