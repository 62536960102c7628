"""Neural point descriptors (mirror of READ/models/texture.py:14-70).

``PointTexture`` keeps the reference's parameter (``texture_`` of shape (1, C, N), same
state-dict key) so checkpoints interchange, and serves lookups from an N x C row-major copy in
HBM through the HIP gather (one 32-byte row per point at C = 8 instead of C reads at stride 4N).
"""
import ctypes as C

import torch
import torch.nn as nn

from . import _lib

_ACT = {"none": 0, "sigmoid": 1, "tanh": 2}


def texture_to_rows(texture_cn):
    """(C, N) channel-major CUDA tensor -> (N, C) row-major CUDA tensor."""
    Cc, n = texture_cn.shape
    rows = torch.empty((n, Cc), dtype=torch.float32, device=texture_cn.device)
    _lib.check(_lib.lib().read_texture_to_rows(texture_cn.data_ptr(), n, Cc, rows.data_ptr(), _lib.stream_ptr()),
               "read_texture_to_rows")
    return rows


def rows_to_texture(rows_nc):
    n, Cc = rows_nc.shape
    tex = torch.empty((Cc, n), dtype=torch.float32, device=rows_nc.device)
    _lib.check(_lib.lib().read_rows_to_texture(rows_nc.data_ptr(), n, Cc, tex.data_ptr(), _lib.stream_ptr()),
               "read_rows_to_texture")
    return tex


def gather_pyramid(rows_nc, idx_levels, activation="none", out=None, ss=1):
    """rows (N,C) + int32 index maps [(B,h,w)...] -> NHWC feature maps [(B,h,w,C)...].
    ss > 1: the index maps were rendered at ss x the feature size (READ/gl/nn.py:100-101) and the samples are reduced by
    the bilinear downscale of READ/models/compose.py:162-163 inside the same launch -> [(B,h/ss,w/ss,C)...]."""
    n, Cc = rows_nc.shape
    levels = len(idx_levels)
    if ss > 1:
        B = int(idx_levels[0].shape[0])
        hs = [int(i.shape[1]) // ss for i in idx_levels]
        wsz = [int(i.shape[2]) // ss for i in idx_levels]
        for i, h, w in zip(idx_levels, hs, wsz):
            if i.shape[0] != B or h * ss != i.shape[1] or w * ss != i.shape[2]:
                raise ValueError(f"index map {tuple(i.shape)} is not a multiple of supersampling {ss}")
        if out is None:
            out = [torch.empty((B, h, w, Cc), dtype=torch.float32, device=rows_nc.device) for h, w in zip(hs, wsz)]
        _lib.check(_lib.lib().read_gather_forward_ss(
            rows_nc.data_ptr(), n, Cc, levels, B, _lib.ptr_array([i.data_ptr() for i in idx_levels]),
            (C.c_int * levels)(*hs), (C.c_int * levels)(*wsz), int(ss), _lib.ptr_array([o.data_ptr() for o in out]),
            _ACT[activation], _lib.stream_ptr()), "read_gather_forward_ss")
        return out
    if out is None:
        out = [torch.empty(tuple(i.shape) + (Cc,), dtype=torch.float32, device=rows_nc.device) for i in idx_levels]
    counts = (C.c_int64 * levels)(*[int(i.numel()) for i in idx_levels])
    _lib.check(_lib.lib().read_gather_forward(
        rows_nc.data_ptr(), n, Cc, levels, _lib.ptr_array([i.data_ptr() for i in idx_levels]), counts,
        _lib.ptr_array([o.data_ptr() for o in out]), _ACT[activation], _lib.stream_ptr()), "read_gather_forward")
    return out


def scatter_pyramid(dfeat_levels, idx_levels, n):
    """Backward of gather_pyramid: -> (N, C) gradient rows (fp32 atomics)."""
    Cc = dfeat_levels[0].shape[-1]
    drows = torch.zeros((n, Cc), dtype=torch.float32, device=dfeat_levels[0].device)
    levels = len(idx_levels)
    dfeat_levels = [d.contiguous() for d in dfeat_levels]
    counts = (C.c_int64 * levels)(*[int(i.numel()) for i in idx_levels])
    _lib.check(_lib.lib().read_gather_backward(
        drows.data_ptr(), n, Cc, levels, _lib.ptr_array([i.data_ptr() for i in idx_levels]), counts,
        _lib.ptr_array([d.data_ptr() for d in dfeat_levels]), _lib.stream_ptr()), "read_gather_backward")
    return drows


class _BilinearDownFn(torch.autograd.Function):
    """(B,C,ss*h,ss*w) -> (B,C,h,w): F.interpolate(scale_factor=1/ss, mode='bilinear') (READ/models/compose.py:162-163)."""

    @staticmethod
    def forward(ctx, x, ss):
        x = x.contiguous()
        B, c, H, W = (int(v) for v in x.shape)
        if H % ss or W % ss:
            raise ValueError(f"input {W}x{H} is not a multiple of supersampling {ss}")
        out = torch.empty((B, c, H // ss, W // ss), dtype=torch.float32, device=x.device)
        _lib.check(_lib.lib().read_bilinear_down(x.data_ptr(), B * c, H // ss, W // ss, int(ss), out.data_ptr(), _lib.stream_ptr()),
                   "read_bilinear_down")
        ctx.cfg = (B, c, H, W, int(ss))
        return out

    @staticmethod
    def backward(ctx, g):
        B, c, H, W, ss = ctx.cfg
        g = g.contiguous()
        din = torch.empty((B, c, H, W), dtype=torch.float32, device=g.device)
        _lib.check(_lib.lib().read_bilinear_down_backward(g.data_ptr(), B * c, H // ss, W // ss, ss, din.data_ptr(),
                                                          _lib.stream_ptr()), "read_bilinear_down_backward")
        return din, None


def bilinear_down(x, ss):
    """HIP replacement of ``F.interpolate(x, scale_factor=1/ss, mode='bilinear')`` for integer ss >= 2 (differentiable)."""
    _lib.require_gpu()
    if not x.is_cuda:
        raise _lib.ReadHipError("bilinear_down runs on the GPU")
    return _BilinearDownFn.apply(x.float(), int(ss))


class _RangeCheck:
    """Out-of-range point ids without a device->host sync per lookup.  The reference's index_select (texture.py:61) reports
    them through an asynchronous device assert; here every lookup queues `max(ids) >= n` into pinned host memory and the
    NEXT lookup (or ``flush()``) raises IndexError once that copy has landed — the gather itself clamps, so nothing reads
    out of bounds in the meantime."""

    def __init__(self):
        self.pending = []          # (event, pinned flag tensor, n)
        self.free = []             # pinned one-int buffers whose copy has landed

    def queue(self, ids, n, signed=False):
        """signed: ids may also be negative (ids that did not come from the rasteriser) — a negative id counts as id n."""
        if ids.numel() == 0:                               # an empty lookup has nothing out of range (and no .max())
            return
        flag = self.free.pop() if self.free else torch.empty(1, dtype=torch.int32).pin_memory()
        worst = (torch.where(ids < 0, n, ids) if signed else ids).max()
        flag.copy_(worst.to(torch.int32).reshape(1), non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self.pending.append((ev, flag, n))

    def poll(self, wait=False):
        keep = []
        for ev, flag, n in self.pending:
            if wait:
                ev.synchronize()
            if ev.query():
                worst = int(flag[0])
                self.free.append(flag)
                if worst >= n:
                    self.pending = []
                    raise IndexError(f"point id {worst} out of range for a descriptor table of {n} points "
                                     f"(wrong texture for this scene?)")
            else:
                keep.append((ev, flag, n))
        self.pending = keep

    def flush(self):
        self.poll(wait=True)


class _GatherFn(torch.autograd.Function):
    """texture_ (1,C,N) x int32 ids (B,H,W) -> NHWC (B,H,W,C); backward = HIP scatter-add."""

    @staticmethod
    def forward(ctx, texture, ids):
        rows = texture_to_rows(texture[0])
        ctx.save_for_backward(ids)
        ctx.n = texture.shape[-1]
        return gather_pyramid(rows, [ids])[0]

    @staticmethod
    def backward(ctx, grad):
        (ids,) = ctx.saved_tensors
        drows = scatter_pyramid([grad], [ids], ctx.n)
        return rows_to_texture(drows)[None], None


class _GatherRowsFn(torch.autograd.Function):
    """Sparse-training lookup: the live (N,C) rows are the master copy.  Backward only QUEUES its (ids, gradient rows) pair on
    the texture (``take_pending``): SparseDescriptorRMSprop sorts the step's pairs by id, sums runs of equal ids and updates
    each touched row once — no N x C gradient table is written (scatter-adding into one cost 10.5 ms per iteration at
    30 M points: random fp32 atomics into 320 MB).  ``grad_rows()`` still materialises the dense gradient rows on request
    (then the optimizer takes them plus the touched ids).  ``texture_`` itself gets no dense (1,C,N) gradient — at 30 M
    points that tensor alone is 960 MB per step."""

    @staticmethod
    def forward(ctx, texture, ids, module):
        ctx.module = module
        ctx.save_for_backward(ids)
        return gather_pyramid(module.training_rows(), [ids])[0]

    @staticmethod
    def backward(ctx, grad):
        (ids,) = ctx.saved_tensors
        m = ctx.module
        g = grad.contiguous()
        m._pending.append((ids.reshape(-1), g.reshape(-1, g.shape[-1])))
        ev = torch.cuda.Event()                     # backward may run on a batch item's own stream (NetAndTexture.forward)
        ev.record()
        m._scatter_events.append(ev)
        return None, None, None


class Texture(nn.Module):
    def null_grad(self):
        raise NotImplementedError()

    def reg_loss(self):
        return 0.


class PointTexture(Texture):
    """Same constructor, parameter name/shape and ``forward`` contract as the reference."""

    def __init__(self, num_channels, size, activation='none', checkpoint=None, init_method='zeros', reg_weight=0.):
        super().__init__()
        assert isinstance(size, int), 'size must be int'
        shape = 1, num_channels, size
        if checkpoint:
            self.texture_ = torch.load(checkpoint, map_location='cpu')['texture'].texture_
        else:
            if init_method == 'rand':
                texture = torch.rand(shape)
            elif init_method == 'zeros':
                texture = torch.zeros(shape)
            else:
                raise ValueError(init_method)
            self.texture_ = nn.Parameter(texture.float())
        if activation not in _ACT:
            raise ValueError(activation)
        self.activation = activation
        self.reg_weight = reg_weight
        self._rows = None
        self._rows_version = None
        self.sparse_training = False        # True: gradients go to grad_rows() + touched ids (SparseDescriptorRMSprop)
        self._range_check = _RangeCheck()   # asynchronous id range check of forward() (raises at the next lookup / check_ids())
        self._grad_rows = None
        self._touched = []
        self._pending = []                  # (ids, gradient rows) pairs of backward passes the optimizer has not consumed
        self._scatter_events = []
        self._rows_newer = False            # the rows were stepped by the sparse optimizer; texture_ is stale until synced

    def null_grad(self):
        self.texture_.grad = None
        self._touched = []
        self._pending = []
        self._scatter_events = []

    # ---- sparse training state ---------------------------------------------------------------------------------------
    def training_rows(self):
        return self.rows()

    def _join_streams(self):
        cur = torch.cuda.current_stream()
        for ev in self._scatter_events:             # gradients produced on other streams must have landed
            cur.wait_event(ev)
        self._scatter_events = []

    def grad_rows(self):
        """Dense (N,C) gradient rows (zero where no pixel pointed).  Queued pairs are scatter-added on the way."""
        rows = self.rows()
        if self._grad_rows is None or self._grad_rows.shape != rows.shape or self._grad_rows.device != rows.device:
            self._grad_rows = torch.zeros_like(rows)
        if self._pending:
            self._join_streams()
            drows = self._grad_rows
            for ids, g in self._pending:
                counts = (C.c_int64 * 1)(int(ids.numel()))
                _lib.check(_lib.lib().read_gather_backward(drows.data_ptr(), drows.shape[0], drows.shape[1], 1,
                                                           _lib.ptr_array([ids.data_ptr()]), counts,
                                                           _lib.ptr_array([g.data_ptr()]), _lib.stream_ptr()),
                           "read_gather_backward")
                self._touched.append(ids)
            self._pending = []
        return self._grad_rows

    def take_pending(self):
        """(ids (n,) int32, gradient rows (n,C)) of every backward pass since the last optimizer step, or None."""
        if not self._pending:
            return None
        self._join_streams()
        ids = torch.cat([p[0] for p in self._pending]).contiguous()
        g = torch.cat([p[1] for p in self._pending]).contiguous()
        self._pending = []
        return ids, g

    def take_touched(self):
        if not self._touched:
            return None
        self._join_streams()
        ids = torch.cat(self._touched).contiguous()
        self._touched = []
        return ids

    def rows_changed(self):
        self._rows_newer = True

    def sync_texture(self):
        """Write the (N,C) rows back into ``texture_`` (checkpoints, dense consumers) after sparse optimizer steps."""
        if self._rows_newer and self._rows is not None:
            self.texture_.data.copy_(rows_to_texture(self._rows)[None])     # .data: no version bump, the row cache stays valid
            self._rows_newer = False

    def state_dict(self, *args, **kwargs):
        self.sync_texture()
        return super().state_dict(*args, **kwargs)

    def _apply(self, fn, *args, **kwargs):
        """``.cpu()`` / ``.cuda()`` / ``.to()`` (NetAndTexture.unload_textures, READ/models/compose.py:118-123; the train loop
        moves the texture off the device between train and eval, train.py:271-305): rows stepped by the sparse optimizer are
        written back into ``texture_`` BEFORE the parameter leaves the device, and the row cache is dropped — the next lookup
        rebuilds it from the (now current) parameter wherever it lives."""
        self.sync_texture()
        out = super()._apply(fn, *args, **kwargs)
        self._rows = None
        self._rows_version = None
        self._rows_newer = False
        self._grad_rows = None
        return out

    def invalidate(self):
        """Drop the cached rows (call after editing ``texture_`` through ``.data`` or other version-blind paths)."""
        self._rows = None
        self._rows_newer = False

    def reg_loss(self):
        return self.reg_weight * torch.mean(torch.pow(self.texture_, 2))

    def rows(self):
        """Cached (N, C) row-major copy of texture_ on its device; refreshed when the parameter changes."""
        t = self.texture_
        key = (t._version, t.data_ptr(), t.device)
        if self._rows is None or self._rows_version != key:
            _lib.require_gpu()
            if not t.is_cuda:
                raise _lib.ReadHipError("PointTexture lookups run on the GPU: move the module with .cuda()")
            self._rows = texture_to_rows(t.detach()[0].contiguous())
            self._rows_version = key
        return self._rows

    @staticmethod
    def _ids(inputs):
        if isinstance(inputs, dict):
            ids = None
            for f, x in inputs.items():
                if 'uv' in f:
                    ids = x[:, 0]
            assert ids is not None, 'Input format does not have uv'
        else:
            ids = inputs[:, 0]                      # B x H x W
        if ids.dtype != torch.int32:
            ids = ids.to(torch.int32)               # float ids are integer valued (texture.py:53 .long())
        return ids.contiguous()

    def forward(self, inputs):
        """ids (B,1|3,H,W) float or int -> (B,C,H,W) view of an NHWC tensor (values as texture.py:55-63)."""
        ids = self._ids(inputs)
        if not self.texture_.is_cuda:
            raise _lib.ReadHipError("PointTexture lookups run on the GPU: move the module with .cuda()")
        ids = ids.to(self.texture_.device)
        # out-of-range ids are reported one lookup late (or at check_ids()): two int(ids.max()) / int(ids.min()) round trips per
        # lookup were 80 device -> host synchronisations per training iteration (8 items x 5 levels); the kernels clamp
        self._range_check.poll()
        self._range_check.queue(ids, self.texture_.shape[-1], signed=True)
        if torch.is_grad_enabled() and self.texture_.requires_grad:
            sample = (_GatherRowsFn.apply(self.texture_, ids, self) if self.sparse_training
                      else _GatherFn.apply(self.texture_, ids))
            if self.activation == 'sigmoid':
                sample = torch.sigmoid(sample)
            elif self.activation == 'tanh':
                sample = torch.tanh(sample)
        else:
            sample = gather_pyramid(self.rows(), [ids], self.activation)[0]
        return sample.permute(0, 3, 1, 2)

    def forward_pyramid(self, idx_levels):
        """Fast path: int32 index maps of all scales -> NHWC feature maps, one launch."""
        return gather_pyramid(self.rows(), idx_levels, self.activation)

    def check_ids(self):
        """Wait for the queued id range checks of forward() and raise IndexError if one of them failed."""
        self._range_check.flush()
