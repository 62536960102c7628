"""Device-resident multi-scale point rasteriser (front end of ``read_splat_forward``).

Replaces the reference's per-call upload + 5 x B kernel launches + download
(MyRender/CloudProjection/pcpr_cuda.cpp:23-42, point_render.cu:169-200,
src/READ/gl/myrender.py:32-40) with a cloud that stays in HBM, one pass over it per frame
for all cameras and all scales, and outputs that stay on the device.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from .camera import level_sizes


class PointCloudRasterizer:
    """Holds xyz (N,3) fp32 in HBM plus the persistent 64-bit key image.

    ``render(total_m, W, H, levels)`` -> (idx_levels, depth_levels): lists of int32 / fp32 CUDA
    tensors shaped (B, h_l, w_l).  Deterministic: per pixel min depth, ties -> min point id;
    empty pixels are (0, 0.0)."""

    CELLS_MIN_POINTS = 1 << 20      # below this the plain pass is used (read_splat_forward_cells falls back anyway)

    def __init__(self, xyz, device=None, cells=True):
        """cells: True = build the cell-ordered copy here (host, multi-threaded); False = plain path only; a uint8
        CUDA tensor = a blob built elsewhere (e.g. by rank 0 and broadcast over RCCL, read_amd/sweep.py)."""
        self.device = device if device is not None else _lib.require_gpu()
        xyz = torch.as_tensor(np.ascontiguousarray(xyz, dtype=np.float32) if not torch.is_tensor(xyz) else xyz)
        if xyz.dim() != 2 or xyz.shape[1] != 3:
            raise ValueError(f"xyz must be (N,3), got {tuple(xyz.shape)}")
        self.xyz = xyz.to(device=self.device, dtype=torch.float32).contiguous()
        self.n = int(self.xyz.shape[0])
        self._workspaces = {}
        self._ws = None
        # cell-ordered copy (Morton-sorted chunks of 1024 points + bounding boxes), built once on the host
        self.cells = None
        if torch.is_tensor(cells):
            if cells.dtype != torch.uint8 or cells.numel() != _lib.lib().read_splat_cells_bytes(self.n):
                raise ValueError("cells blob does not belong to a cloud of this size")
            self.cells = cells.to(self.device)
        elif cells and self.n >= self.CELLS_MIN_POINTS:
            self.cells = torch.from_numpy(build_cells(xyz.detach().cpu().numpy())).to(self.device)
        if self.cells is not None:           # a fresh blob at an address the allocator may have handed out before
            _lib.check(_lib.lib().read_splat_cells_invalidate(self.cells.data_ptr(), self.n), "read_splat_cells_invalidate")

    def _workspace(self, B, W, H):
        """One persistent workspace per (min(B,8), W, H): key images, hi-z bounds and the previous frame's
        winners (the warm start of the next frame rendered at that size)."""
        key = (min(B, 8), W, H)
        ws = self._workspaces.get(key)
        if ws is None:
            need = _lib.lib().read_splat_workspace_bytes(B, W, H)
            ws = torch.empty(need, dtype=torch.uint8, device=self.device)
            _lib.check(_lib.lib().read_splat_workspace_init(ws.data_ptr(), ws.numel(), _lib.stream_ptr()),
                       "read_splat_workspace_init")
            self._workspaces[key] = ws
        self._ws = ws
        return ws

    def render(self, total_m, W, H, levels=5, want_depth=True, out=None, next_total=None):
        """total_m: (B,4,4) or (4,4) fp32 host array/tensor = proj @ inv(view).
        next_total: the matrix the NEXT call at this size will use, when the caller knows it (a sweep, a trajectory replay): this
        frame's last launch then also prepares the next frame's chunk lists and depth seeds (read_splat_hint_next_camera —
        4 dependent launches per frame instead of 5; results identical, a wrong announcement only costs two small memsets)."""
        M = np.ascontiguousarray(total_m.detach().cpu().numpy() if torch.is_tensor(total_m) else total_m,
                                 dtype=np.float32).reshape(-1, 16)
        B = M.shape[0]
        sizes = level_sizes(W, H, levels)
        if out is None:
            idx = [torch.empty((B, h, w), dtype=torch.int32, device=self.device) for (w, h) in sizes]
            dep = [torch.empty((B, h, w), dtype=torch.float32, device=self.device) for (w, h) in sizes] \
                if want_depth else None
        else:
            idx, dep = out
        ws = self._workspace(B, W, H)
        L = _lib.lib()
        if next_total is not None and B == 1 and self.cells is not None:
            Mn = np.ascontiguousarray(next_total.detach().cpu().numpy() if torch.is_tensor(next_total) else next_total,
                                      dtype=np.float32).reshape(-1, 16)
            _lib.check(L.read_splat_hint_next_camera(ws.data_ptr(), Mn.ctypes.data_as(C.POINTER(C.c_float))),
                       "read_splat_hint_next_camera")
        idx_p = _lib.ptr_array([t.data_ptr() for t in idx])
        dep_p = _lib.ptr_array([t.data_ptr() for t in dep]) if dep is not None else None
        _lib.check(L.read_splat_forward_cells(self.xyz.data_ptr(),
                                              self.cells.data_ptr() if self.cells is not None else None, self.n,
                                              M.ctypes.data_as(C.POINTER(C.c_float)), B, W, H, levels, idx_p, dep_p,
                                              ws.data_ptr(), ws.numel(), _lib.stream_ptr()), "read_splat_forward_cells")
        return idx, dep

    def bind(self, W, H, levels, out, totals):
        """A pre-bound frame call for loops that must not be host-bound (benchmark stages): every ctypes argument is built ONCE —
        the outputs ``out`` = (idx levels, depth levels or None) and the list of camera matrices ``totals`` — and
        ``call(k, next_k=None)`` then costs two foreign calls (hint + forward), ~5 us of host time instead of the ~80 us of
        ``render()``'s argument marshalling.  Same C entry points, same results."""
        idx, dep = out
        ws = self._workspace(1, W, H)
        L = _lib.lib()
        Ms = [np.ascontiguousarray(t.detach().cpu().numpy() if torch.is_tensor(t) else t, dtype=np.float32).reshape(16) for t in totals]
        Mp = [m.ctypes.data_as(C.POINTER(C.c_float)) for m in Ms]
        idx_p = _lib.ptr_array([t.data_ptr() for t in idx])
        dep_p = _lib.ptr_array([t.data_ptr() for t in dep]) if dep is not None else None
        xyz_p, cells_p, ws_p, ws_n = self.xyz.data_ptr(), self.cells.data_ptr() if self.cells is not None else None, ws.data_ptr(), ws.numel()
        fwd, hint, n, check = L.read_splat_forward_cells, L.read_splat_hint_next_camera, self.n, _lib.check
        keep = (Ms, idx, dep, ws)

        def call(k, next_k=None, stream=None):
            st = _lib.stream_ptr() if stream is None else stream
            if next_k is not None and cells_p is not None:
                hint(ws_p, Mp[next_k])
            check(fwd(xyz_p, cells_p, n, Mp[k], 1, W, H, levels, idx_p, dep_p, ws_p, ws_n, st), "read_splat_forward_cells")
        call.keep = keep
        return call

    def render_gl(self, total_m, W, H, point_size=1.0, relative=False, min_point_size=1.0, discard=None, drop=None,
                  perturb=None, perturb_hash=None, want_depth=True, point_sizes=None):
        """ONE level of ONE camera at its own size with the GL twin's point options (read_splat_forward_gl):
        point_size / relative ("pN" / "psN" tokens, READ/gl/programs.py:183-192), discard = bool/uint8 (N,) array or
        tensor (set_point_discard), drop = (p, seed) seeded drop, perturb = (N,2) clip-space offsets
        (set_point_perturb), perturb_hash = (amp, seed), point_sizes = (N,) per-point sizes (set_point_sizes; they replace
        point_size as in the shader, programs.py:183-187).  -> (idx (1,H,W) int32, depth (1,H,W) fp32 | None)."""
        M = np.ascontiguousarray(total_m.detach().cpu().numpy() if torch.is_tensor(total_m) else total_m,
                                 dtype=np.float32).reshape(-1, 16)
        if M.shape[0] != 1:
            raise ValueError("render_gl renders one camera per call")
        o = _lib.SplatGlOpts(float(point_size), int(bool(relative)), float(min_point_size), None, 0, 0, None, 0.0, 0, None)
        keep = []
        if point_sizes is not None:         # per-point sizes (NNScene.set_point_sizes): they replace the token's global size
            ps = torch.as_tensor(point_sizes).to(self.device, torch.float32).contiguous().reshape(-1)
            if ps.numel() != self.n:
                raise ValueError(f"point_sizes has {ps.numel()} entries for {self.n} points")
            keep.append(ps)
            o.point_sizes = ps.data_ptr()
            o.point_size = 0.0
        if discard is not None:
            d = torch.as_tensor(discard).to(self.device, torch.uint8).contiguous()
            if d.numel() != self.n:
                raise ValueError(f"discard mask has {d.numel()} entries for {self.n} points")
            keep.append(d)
            o.discard = d.data_ptr()
        if drop is not None:
            o.drop_threshold, o.drop_seed = drop_threshold(drop[0]), int(drop[1]) & 0xffffffff
        if perturb is not None:
            pt = torch.as_tensor(perturb).to(self.device, torch.float32).contiguous()
            if tuple(pt.shape) != (self.n, 2):
                raise ValueError(f"perturb must be ({self.n}, 2), got {tuple(pt.shape)}")
            keep.append(pt)
            o.perturb = pt.data_ptr()
        if perturb_hash is not None:
            o.perturb_amp, o.perturb_seed = float(perturb_hash[0]), int(perturb_hash[1]) & 0xffffffff
        idx = torch.empty((1, H, W), dtype=torch.int32, device=self.device)
        dep = torch.empty((1, H, W), dtype=torch.float32, device=self.device) if want_depth else None
        ws = self._workspace(1, W, H)
        _lib.check(_lib.lib().read_splat_forward_gl(self.xyz.data_ptr(), self.n, M.ctypes.data_as(C.POINTER(C.c_float)), W, H,
                                                    C.byref(o), idx.data_ptr(), dep.data_ptr() if dep is not None else None,
                                                    ws.data_ptr(), ws.numel(), _lib.stream_ptr()), "read_splat_forward_gl")
        return idx, dep


def drop_threshold(p):
    """Probability -> u32 threshold of the seeded drop (point i dropped iff rnd(i, seed) < threshold)."""
    return int(min(max(float(p), 0.0), 1.0) * 4294967295.0)


def build_cells(xyz):
    """Host blob of ``read_splat_cells_build_host`` for an (N,3) float32 array (uint8 ndarray, upload as is)."""
    xyz = np.ascontiguousarray(xyz, dtype=np.float32)
    L = _lib.lib()
    nbytes = L.read_splat_cells_bytes(xyz.shape[0])
    if nbytes == 0:
        raise ValueError(f"cannot build cells for {xyz.shape[0]} points")
    blob = np.empty(nbytes, np.uint8)
    _lib.check(L.read_splat_cells_build_host(xyz.ctypes.data, xyz.shape[0], blob.ctypes.data, nbytes),
               "read_splat_cells_build_host")
    return blob


def index_to_float(idx):
    """The reference's float32 index image (point_render.cu:158): ids >= 2**24 round."""
    out = torch.empty(idx.shape, dtype=torch.float32, device=idx.device)
    _lib.check(_lib.lib().read_index_to_float(idx.data_ptr(), idx.numel(), out.data_ptr(), _lib.stream_ptr()),
               "read_index_to_float")
    return out
