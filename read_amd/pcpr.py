"""Drop-in for the reference's ``pcpr`` torch extension (MyRender/CloudProjection/pcpr_cuda.cpp:23-42).

    import read_amd.pcpr as pcpr
    index, depth = pcpr.forward(points, total_m, w, h, 512)

Same signature, same return (two fresh CPU float32 tensors (B,H,W): float point index and depth,
empty = (0, 0)), same validation errors (RuntimeError on non-float / non-contiguous inputs or a
total_m that is not 3-D).  Differences, all on the side of the caller: the cloud is uploaded once
and cached by tensor identity instead of on every call (pcpr_cuda.cpp:29), the result is
deterministic (min depth, ties -> min id) instead of arrival-order dependent, and `block_size`
is accepted and ignored (a launch-shape hint of the CUDA kernel).
"""
import weakref

import torch

from . import _lib
from .raster import PointCloudRasterizer, index_to_float

_CACHE = {}          # id(tensor) -> (weakref, version, data_ptr, rasterizer)
_CACHE_MAX = 8


def _rasterizer_for(points):
    key = id(points)
    hit = _CACHE.get(key)
    if hit is not None:
        ref, version, ptr, r = hit
        if ref() is points and version == points._version and ptr == points.data_ptr():
            return r
    r = PointCloudRasterizer(points)
    if len(_CACHE) >= _CACHE_MAX:
        _CACHE.pop(next(iter(_CACHE)))
    _CACHE[key] = (weakref.ref(points), points._version, points.data_ptr(), r)
    return r


def clear_cache():
    _CACHE.clear()


def forward(in_points, total_m, tar_width, tar_height, block_size=512):
    if not (torch.is_tensor(in_points) and torch.is_tensor(total_m)):
        raise RuntimeError("in_points and total_m must be tensors")
    for name, t in (("in_points", in_points), ("total_m", total_m)):
        if not t.is_contiguous():
            raise RuntimeError(f"{name} must be contiguous")
        if t.dtype != torch.float32:
            raise RuntimeError(f"{name} must be a float tensor")
    if total_m.dim() != 3:
        raise RuntimeError("batch_size check")
    _lib.require_gpu()
    r = _rasterizer_for(in_points)
    idx, dep = r.render(total_m, int(tar_width), int(tar_height), levels=1)
    out_index = index_to_float(idx[0]).cpu()
    out_depth = dep[0].cpu()
    return [out_index, out_depth]
