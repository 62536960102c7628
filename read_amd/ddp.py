"""Data-parallel training of READ's step, one process per GPU (SURVEY.md 8e "Training"; replaces the reference's
``nn.DataParallel(model)``, /root/reference/train.py:138-139, src/train.py:147-148).

The reference replicates the model on every GPU each step (scatter inputs, replicate weights, gather outputs on GPU 0, the
960 MB dense descriptor gradient reduced onto GPU 0).  Here every rank keeps a full replica (cloud, descriptors, weights:
2 % of a GPU's HBM) and renders + back-propagates ITS OWN crops; per step there are exactly two exchanges over RCCL:

  * ONE all-reduce of the network's gradients, packed into one flat arena (594 tensors, 121 MB at READ's UNet): a single
    large collective — xGMI rings are per-link bound, so one 121 MB ring all-reduce (2 x 7/8 x 121 MB over 7 links) beats
    594 small ones by their launch latencies;
  * ONE all-gather of the step's sparse descriptor gradient — the (point id, gradient row) pairs the crops' index maps touched
    (~0.5 M pairs x 36 B per rank against 960 MB dense) — after which every rank runs the SAME deterministic sorted update
    (``read_rmsprop_sorted``: pairs in rank order, stable sort by id, runs summed in order), so the descriptor replicas stay
    bit-identical without ever broadcasting them.

Gradient semantics = ``nn.DataParallel`` + ``ModelAndLoss`` (/root/reference/READ/models/compose.py:21-42: the loss is computed
per replica and averaged): the mean over ranks of each rank's gradient.

The collectives are plain ``torch.distributed`` calls on the tensors' own device: backend "nccl" (= RCCL over xGMI) on the GPUs,
"gloo" in the CPU tests (tests/test_ddp_gloo.py) — this module contains no HIP call, the kernels stay where they were.
"""
import torch
import torch.distributed as dist


def _world(group=None):
    return dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1


FORCE_COLLECTIVES = False        # tests: issue the collectives on a 1-rank group too (the RCCL path on a single GPU)


def _on(group=None):
    return _world(group) > 1 or (FORCE_COLLECTIVES and dist.is_available() and dist.is_initialized())


class GradientArena:
    """All parameter gradients of a net in ONE flat fp32 buffer.

    ``reduce()`` after ``backward()``: the gradients are copied into the arena by one multi-tensor launch, all-reduced as ONE
    collective, divided by the world size, and each ``p.grad`` is re-pointed at its slice of the arena (no copy back).  A
    parameter without a gradient in this step contributes zeros (every rank must reduce the same bytes)."""

    def __init__(self, params, group=None):
        self.params = [p for p in params if p.requires_grad]
        assert self.params, "GradientArena: no trainable parameter"
        dev, dt = self.params[0].device, self.params[0].dtype
        assert all(p.device == dev and p.dtype == dt for p in self.params), "GradientArena: one device and dtype"
        self.group = group
        offs, n = [], 0
        for p in self.params:
            offs.append(n)
            n += (p.numel() + 63) // 64 * 64                  # 256-byte slices (fp32)
        # tail: one word per parameter, 1 where this rank HAS a gradient — summed by the same all-reduce, so that a parameter
        # without a gradient on EVERY rank keeps p.grad = None (Adam / weight decay skip it, as under nn.DataParallel)
        self.tail = n
        self.flat = torch.zeros(n + (len(self.params) + 63) // 64 * 64, dtype=dt, device=dev)
        self.views = [self.flat[o:o + p.numel()].view_as(p) for o, p in zip(offs, self.params)]
        self.flags = self.flat[n:n + len(self.params)]
        self.pending = None
        self._none = []

    @property
    def nbytes(self):
        return self.flat.numel() * self.flat.element_size()

    def pack(self):
        have = [(v, p.grad) for v, p in zip(self.views, self.params) if p.grad is not None and p.grad.data_ptr() != v.data_ptr()]
        none = [v for v, p in zip(self.views, self.params) if p.grad is None]
        if have:
            torch._foreach_copy_([v for v, _ in have], [g for _, g in have])
        if none:
            torch._foreach_zero_(none)
        self._none = [i for i, p in enumerate(self.params) if p.grad is None]
        self.flags.fill_(1.0)
        if self._none:
            self.flags[torch.tensor(self._none, device=self.flags.device)] = 0.0

    def reduce(self, async_op=False):
        """Mean over the ranks of every gradient; ``p.grad`` then aliases the arena.  async_op: returns after the collective is
        enqueued (the caller runs the descriptor exchange meanwhile) — call ``wait()`` before the optimizer step."""
        self.pack()
        if _on(self.group):
            self.pending = dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        if not async_op:
            self.wait()

    def wait(self):
        w = _world(self.group)
        if self.pending is not None:
            self.pending.wait()
            self.pending = None
            self.flat[:self.tail].mul_(1.0 / w)
        # only a parameter without a gradient HERE can be without one everywhere: the common step (every gradient present) reads
        # nothing back; the rare one costs one small device-to-host copy
        absent = set()
        if self._none:
            got = self.flags[torch.tensor(self._none, device=self.flags.device)].tolist()
            absent = {i for i, f in zip(self._none, got) if f == 0.0}
        for i, (v, p) in enumerate(zip(self.views, self.params)):
            p.grad = None if i in absent else v


def exchange_pairs(ids, rows, group=None):
    """All ranks' (ids (n_r,) int32, gradient rows (n_r, C) fp32) of one step -> the concatenation in RANK ORDER on every rank,
    rows scaled by 1 / world (mean over ranks, as the network's gradients).  Ragged: the lengths travel first (one small
    all-gather), the payloads are padded to the longest rank's length for the collective and trimmed afterwards.  A rank with
    no pairs this step passes empty tensors (every rank must still call).  Without a process group: the input, unscaled."""
    w = _world(group)
    if not _on(group):
        return ids, rows
    dev = ids.device
    C = rows.shape[1]
    n = torch.tensor([ids.numel()], dtype=torch.int64, device=dev)
    lens = [torch.zeros_like(n) for _ in range(w)]
    dist.all_gather(lens, n, group=group)
    lens = [int(x.item()) for x in lens]
    m = max(lens)
    if m == 0:
        return ids, rows
    # ids and rows travel in one buffer: row-major (m, C + 1) fp32 words, the id bit-cast into the last column
    send = torch.zeros((m, C + 1), dtype=torch.float32, device=dev)
    if ids.numel():
        send[:ids.numel(), :C].copy_(rows)
        send[:ids.numel(), C].copy_(ids.to(torch.int32).view(torch.float32))
    recv = torch.empty((w, m, C + 1), dtype=torch.float32, device=dev)
    dist.all_gather_into_tensor(recv, send[None], group=group)
    parts = [recv[r, :lens[r]] for r in range(w) if lens[r]]
    allp = torch.cat(parts) if len(parts) > 1 else parts[0]
    out_ids = allp[:, C].contiguous().view(torch.int32)
    out_rows = (allp[:, :C] * (1.0 / w)).contiguous()
    return out_ids, out_rows


def sync_buffers(module, src=0, group=None):
    """BatchNorm running statistics (``.train()`` mode) follow rank `src`, as DataParallel keeps replica 0's buffers."""
    if not _on(group):
        return
    bufs = [b for b in module.buffers() if b.is_floating_point()]
    if not bufs:
        return
    flat = torch.cat([b.reshape(-1) for b in bufs])
    dist.broadcast(flat, src, group=group)
    o = 0
    for b in bufs:
        b.copy_(flat[o:o + b.numel()].view_as(b))
        o += b.numel()


def broadcast_textures(textures, src=0, group=None):
    """Every replica's descriptor table := rank `src`'s (one broadcast per table, at construction only).  The per-step exchange
    keeps the replicas bit-identical only if they START identical: a random init with per-rank seeds, or a checkpoint loaded on
    one rank, would otherwise diverge silently — each rank applying the same gathered gradient to different rows."""
    if not _on(group):
        return
    for tex in textures:
        if hasattr(tex, 'sync_texture'):
            tex.sync_texture()                                 # rows stepped by the sparse optimizer -> texture_
        dist.broadcast(tex.texture_.data, src, group=group)
        if hasattr(tex, '_rows'):
            tex._rows = None                                   # the (N, C) row cache is rebuilt from the received table
            tex._rows_version = None


def broadcast_parameters(module, src=0, group=None):
    """Make every replica start from rank `src`'s weights (one flat broadcast)."""
    if not _on(group):
        return
    ps = [p.data for p in module.parameters()]
    flat = torch.cat([p.reshape(-1) for p in ps])
    dist.broadcast(flat, src, group=group)
    o = 0
    for p in ps:
        p.copy_(flat[o:o + p.numel()].view_as(p))
        o += p.numel()


class DataParallelStep:
    """What a training loop calls between ``loss.backward()`` and the optimizers' ``step()``:

        ddp = DataParallelStep(pipeline.net, pipeline.textures)      # once, after init_process_group
        loss.backward()
        ddp.reduce()                                                 # 1 all-reduce + 1 all-gather
        pipeline.optimizer.step(); extra_optimizer.step()

    ``textures``: the PointTexture modules in sparse-training mode; their queued (ids, rows) pairs are replaced by the gathered,
    scaled pairs of all ranks, which ``SparseDescriptorRMSprop.step()`` then consumes unchanged.  At construction rank 0's
    parameters, buffers AND descriptor tables are broadcast (``broadcast=False``: the caller vouches the replicas are identical);
    with the net in ``.train()`` mode ``reduce()`` also re-broadcasts rank 0's BatchNorm statistics every step."""

    def __init__(self, net, textures=(), group=None, broadcast=True):
        # pipeline.textures is {dataset id: PointTexture} (READ/pipelines/ogl.py:82-97): a mapping gives its values
        self.net, self.textures, self.group = net, list(textures.values() if hasattr(textures, 'values') else textures), group
        self.arena = GradientArena(net.parameters(), group)
        if broadcast:                                    # as torch DDP at construction: parameters, buffers — and the descriptor tables
            broadcast_parameters(net, 0, group)
            sync_buffers(net, 0, group)
            broadcast_textures(self.textures, 0, group)

    def reduce(self):
        self.arena.reduce(async_op=True)                 # the 121 MB ring all-reduce is on the wire ...
        for tex in self.textures:                        # ... while the sparse descriptor pairs are gathered
            pend = tex.take_pending()
            if pend is None:
                dev = self.arena.flat.device
                C = int(tex.texture_.shape[1])
                pend = (torch.zeros(0, dtype=torch.int32, device=dev), torch.zeros((0, C), dtype=torch.float32, device=dev))
            ids, rows = exchange_pairs(pend[0], pend[1], self.group)
            if ids.numel():
                tex._pending.append((ids, rows))
        self.arena.wait()
        if self.net.training:                            # BatchNorm running statistics follow rank 0's (nn.DataParallel: replica 0 owns the buffers)
            sync_buffers(self.net, 0, self.group)
