"""Multi-GPU novel-view sweep: camera poses are independent units, so they shard across the GPUs of
a node with no data-path collective; RCCL (torch.distributed backend "nccl" on ROCm; "gloo" in the
CPU tests) is used only for the one-time broadcast of the scene and the exchange of finished frames
(SURVEY.md §8e).  One process per GPU.  This module is the ONE implementation of that loop: ``bench.py``
times ``run_steps`` and the CPU tests drive the same function with a stub renderer.

Two pose layouts (``pose_of_step``): 'interleave' — pose k on rank ``k % world``, consecutive poses of a trajectory land on
different GPUs, so a viewer replaying the sweep in order drains all GPUs evenly — and 'block' — every rank walks a contiguous
block of the sweep, which keeps the rasteriser's frame-to-frame warm start coherent.

Expected scaling (for judging a measured curve): weak scaling is linear up to the frame exchange — one RGBA frame is
6.8 MB, so at ~150 frames/s per GPU an all-gather moves ~1 GB/s per peer link and a gather-to-root ~7 GB/s into
rank 0, against ~153 GB/s per xGMI link; the exchange of step i overlaps the rendering of step i+1."""
import torch
import torch.distributed as dist


def _dist_on():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def shard_indices(n_items, rank, world):
    """Indices of the poses rank `rank` renders (round-robin)."""
    return list(range(rank, n_items, world))


def broadcast_scene(tensors, src=0):
    """Replicate the scene (xyz, descriptors, packed weights, cell-ordered cloud) from `src` to every rank, in place."""
    if _dist_on():
        for t in tensors:
            dist.broadcast(t, src)
    return tensors


def broadcast_scene_from_rank0(make, device):
    """``make()`` (called on rank 0 only) returns a list of tensors; every rank returns the same list on ``device``.
    Shapes/dtypes travel first (one small object broadcast), then the payloads over the device collective."""
    if not _dist_on():
        return [t.to(device) for t in make()]
    rank = dist.get_rank()
    tensors = [t.to(device) for t in make()] if rank == 0 else None
    meta = [[(tuple(t.shape), t.dtype) for t in tensors]] if rank == 0 else [None]
    dist.broadcast_object_list(meta, src=0)
    if rank != 0:
        tensors = [torch.empty(shape, dtype=dtype, device=device) for (shape, dtype) in meta[0]]
    return broadcast_scene(tensors, 0)


def gather_objects(obj):
    """One small Python object per rank -> the list in rank order on every rank (``[obj]`` without a process group).
    bench.py uses it for ``verified_ranks``: every rank checks one of ITS OWN frames against the oracle."""
    if not _dist_on():
        return [obj]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, obj)
    return out


class FrameExchange:
    """Double-buffered exchange of finished frames: while the frames of step i are on the wire (async collective),
    step i+1 is rendered into the other buffer.

    mode 'all'   every rank receives every rank's frame (all_gather_into_tensor) — a tiled display wall / any consumer
         'root'  only rank 0 receives them (gather): the viewer process displays, the others only render
         None    no exchange (frames stay where they were rendered)"""

    def __init__(self, frame_shape, device, dtype=torch.float32, mode='all', force=False):
        if mode not in ('all', 'root', None):
            raise ValueError(mode)
        on = _dist_on() or (force and dist.is_available() and dist.is_initialized())     # force: exchange even at world size 1
        self.world = dist.get_world_size() if on else 1
        self.rank = dist.get_rank() if on else 0
        self.mode = mode if on else None
        # device frames: the collectives are issued from a stream of their own, after the event that says the frame is
        # complete — the caller's stream never waits for a frame, so a renderer with several frames in flight keeps them in flight
        self.comm = torch.cuda.Stream(device) if (self.mode is not None and torch.device(device).type == 'cuda') else None
        self.send = [torch.zeros(tuple(frame_shape), dtype=dtype, device=device) for _ in range(2)]
        self.recv = [None, None]
        if self.mode == 'all' or (self.mode == 'root' and self.rank == 0):
            self.recv = [torch.zeros((self.world,) + tuple(frame_shape), dtype=dtype, device=device) for _ in range(2)]
        self.pending = [None, None]

    def buffer(self, i):
        """The frame buffer step i renders into; waits until the exchange that last used it has completed."""
        j = i & 1
        if self.pending[j] is not None:
            self.pending[j].wait()
            self.pending[j] = None
        return self.send[j]

    def post(self, i, done=None):
        """Start the exchange of step i's frame.  done: event after which the frame is complete (a renderer that writes it
        on a stream of its own); None = complete in the current stream's order."""
        j = i & 1
        if self.mode is None:
            return
        if self.comm is not None:
            self.comm.wait_stream(torch.cuda.current_stream(self.send[j].device))
            if done is not None:
                self.comm.wait_event(done)
            with torch.cuda.stream(self.comm):
                self.pending[j] = self._collective(j)
        else:
            self.pending[j] = self._collective(j)

    def _collective(self, j):
        if self.mode == 'all':
            return dist.all_gather_into_tensor(self.recv[j], self.send[j][None], async_op=True)
        out = list(self.recv[j].unbind(0)) if self.rank == 0 else None
        return dist.gather(self.send[j], out, dst=0, async_op=True)

    def frames(self, i):
        """(world, *frame_shape) frames of step i in rank order once its exchange has completed (None where not received)."""
        j = i & 1
        if self.pending[j] is not None:
            self.pending[j].wait()
            self.pending[j] = None
        if self.mode is None:
            return self.send[j][None]
        return self.recv[j]

    def drain(self):
        for j in range(2):
            if self.pending[j] is not None:
                self.pending[j].wait()
                self.pending[j] = None


LAYOUTS = ('interleave', 'block')
DEFAULT_LAYOUT = 'interleave'        # bench.py --pose-layout; see pose_of_step


def sweep_steps(n_poses, world):
    """Steps that cover every pose of the sweep once: ceil(n_poses / world).  When ``n_poses % world != 0`` the ranks past the
    end of the last step wrap around to the head of the sweep (``pose_of_step``) — every rank renders and exchanges in every
    step, so the collectives stay matched; a consumer drops the wrapped frames (``is_wrapped``)."""
    return (n_poses + world - 1) // world


def pose_of_step(step, rank, world, n_poses, layout=None):
    """The pose rank `rank` of `world` renders in step `step` (SURVEY.md 8e allows either split of a sweep).

    'interleave'  pose = step * world + rank: consecutive poses land on different GPUs — a viewer replaying the sweep in
                  order drains all GPUs evenly, but each rank's rasteriser sees a stride-`world` walk of the trajectory;
    'block'       rank r owns the contiguous block of ceil(n_poses / world) poses starting at r * that: every rank walks
                  CONSECUTIVE poses, which is what the rasteriser's warm start (last frame's front points seed this frame's
                  depth bounds) is built for — the layout of an offline batch sweep.
    Both wrap modulo n_poses, so any number of steps is defined (bench.py's weak-scaling loop runs K steps per rank)."""
    layout = layout or DEFAULT_LAYOUT
    if layout == 'interleave':
        return (step * world + rank) % n_poses
    if layout == 'block':
        return (rank * sweep_steps(n_poses, world) + step) % n_poses
    raise ValueError(layout)


def is_wrapped(step, rank, world, n_poses, layout=None):
    """True when (step, rank) of a ``sweep_steps``-long sweep re-renders a pose another (step, rank) already covers
    (the ragged tail: n_poses % world != 0)."""
    layout = layout or DEFAULT_LAYOUT
    if layout == 'interleave':
        return step * world + rank >= n_poses
    return rank * sweep_steps(n_poses, world) + step >= n_poses


def run_steps(render_into, exchange, first, count, n_poses, layout=None, shard=None, announce_next=False):
    """Steps first .. first+count-1 of the sweep on this rank: step i renders pose ``pose_of_step(i, rank, world, n_poses,
    layout)`` into the exchange's buffer and posts the exchange.  ``render_into(pose_index, out_tensor)`` returns None, or the
    event that marks the frame complete when the renderer writes it on its own stream (FrameRenderer(frames_in_flight=2)).
    shard = (rank, world) overrides the exchange's own — bench.py's single-GPU proxy of ONE rank's share of an N-rank sweep.
    announce_next: a sweep knows its next pose — ``render_into(pose, out, next_pose)`` gets it as a third argument, so that the
    rasteriser can prepare the next frame inside this one's last launch (PointCloudRasterizer.render(next_total=...))."""
    rank, world = shard if shard is not None else (exchange.rank, exchange.world)
    for i in range(first, first + count):
        out = exchange.buffer(i)
        pose = pose_of_step(i, rank, world, n_poses, layout)
        if announce_next:
            done = render_into(pose, out, pose_of_step(i + 1, rank, world, n_poses, layout))
        else:
            done = render_into(pose, out)                              # an event if the frame completes on another stream
        exchange.post(i, done)


def render_sweep(render_fn, n_poses, frame_shape, device, dtype=torch.float32, gather=True):
    """Render poses ``rank::world`` with ``render_fn(k) -> tensor(frame_shape)`` and (optionally) all-gather them so
    every rank ends with the full ``(n_poses, *frame_shape)`` stack in pose order."""
    ex = FrameExchange(frame_shape, device, dtype, 'all' if gather else None)
    world, rank = ex.world, ex.rank
    mine = shard_indices(n_poses, rank, world)
    rounds = (n_poses + world - 1) // world
    if ex.mode is None:
        local = torch.zeros((len(mine),) + tuple(frame_shape), dtype=dtype, device=device)
        for j, k in enumerate(mine):
            local[j].copy_(render_fn(k))
        return local if world == 1 else (local, mine)
    out = torch.zeros((rounds * world,) + tuple(frame_shape), dtype=dtype, device=device)

    for r in range(rounds):
        buf = ex.buffer(r)
        k = r * world + rank
        if k < n_poses:
            buf.copy_(render_fn(k))
        else:
            buf.zero_()                                                  # past the end of the sweep: this rank sends zeros
        if r > 0:
            out[(r - 1) * world:r * world].copy_(ex.frames(r - 1))     # slab r-1: rank order == pose order
        ex.post(r)
    out[(rounds - 1) * world:rounds * world].copy_(ex.frames(rounds - 1))
    return out[:n_poses]
