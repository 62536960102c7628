"""Multi-GPU novel-view sweep: camera poses are independent units, so they shard across the GPUs of
a node with no data-path collective; RCCL (torch.distributed backend "nccl" on ROCm; "gloo" in the
CPU tests) is used only for the one-time broadcast of the scene and the all-gather of finished
frames.  One process per GPU.

Pose k is rendered by rank ``k % world`` — consecutive poses of a trajectory land on different
GPUs, so a viewer replaying the sweep in order drains all GPUs evenly."""
import torch
import torch.distributed as dist


def shard_indices(n_items, rank, world):
    """Indices of the poses rank `rank` renders (round-robin)."""
    return list(range(rank, n_items, world))


def broadcast_scene(tensors, src=0):
    """Replicate the scene (xyz, descriptors, packed weights) from `src` to every rank, in place."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        for t in tensors:
            dist.broadcast(t, src)
    return tensors


def render_sweep(render_fn, n_poses, frame_shape, device, dtype=torch.float32, gather=True):
    """Render poses ``rank::world`` with ``render_fn(k) -> tensor(frame_shape)`` and (optionally)
    all-gather them so every rank ends with the full ``(n_poses, *frame_shape)`` stack in pose order.

    Frames are exchanged in slabs of one frame per rank: while slab s is on the wire (async
    all-gather), slab s+1 is being rendered."""
    distributed = dist.is_available() and dist.is_initialized()
    world = dist.get_world_size() if distributed else 1
    rank = dist.get_rank() if distributed else 0
    mine = shard_indices(n_poses, rank, world)
    rounds = (n_poses + world - 1) // world
    if not gather or world == 1:
        local = torch.zeros((len(mine),) + tuple(frame_shape), dtype=dtype, device=device)
        for j, k in enumerate(mine):
            local[j].copy_(render_fn(k))
        if world == 1:
            return local
        return local, mine
    out = torch.zeros((rounds * world,) + tuple(frame_shape), dtype=dtype, device=device)
    send = [torch.zeros(tuple(frame_shape), dtype=dtype, device=device) for _ in range(2)]
    pending = [None, None]
    for r in range(rounds):
        j = r & 1
        if pending[j] is not None:
            pending[j].wait()
        k = r * world + rank
        if k < n_poses:
            send[j].copy_(render_fn(k))
        else:
            send[j].zero_()
        # slab r holds poses r*world .. r*world+world-1, i.e. rank order == pose order
        pending[j] = dist.all_gather_into_tensor(out[r * world:(r + 1) * world], send[j][None], async_op=True)
    for p in pending:
        if p is not None:
            p.wait()
    return out[:n_poses]
