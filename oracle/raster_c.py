"""ctypes front-end of oracle/raster.c (ORACLE — test infrastructure only)."""
import ctypes as C
import os

import numpy as np

from . import build as _build

_LIB = None


def lib_path() -> str:
    return _build.OUT


def _lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(_build.OUT):
            _build.build()
        L = C.CDLL(_build.OUT)
        f32p, i32p = C.POINTER(C.c_float), C.POINTER(C.c_int32)
        L.oracle_raster_level.argtypes = [f32p, C.c_int64, f32p, C.c_int, C.c_int, i32p, f32p]
        L.oracle_raster_level_mt.argtypes = [f32p, C.c_int64, f32p, C.c_int, C.c_int, i32p, f32p, C.c_int]
        L.oracle_raster_multiscale.argtypes = [f32p, C.c_int64, f32p, C.c_int, C.c_int, C.c_int,
                                               i32p, f32p, C.c_int]
        L.oracle_index_to_float.argtypes = [i32p, C.c_size_t, f32p]
        L.oracle_gather_chw.argtypes = [f32p, C.c_int64, C.c_int, i32p, C.c_size_t, f32p]
        L.oracle_gather_backward_chw.argtypes = [f32p, i32p, C.c_size_t, C.c_int, C.c_int64, f32p]
        for fn in (L.oracle_raster_level, L.oracle_raster_level_mt, L.oracle_raster_multiscale,
                   L.oracle_index_to_float, L.oracle_gather_chw, L.oracle_gather_backward_chw):
            fn.restype = None
        _LIB = L
    return _LIB


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def level_sizes(W, H, levels):
    """myrender.py:33-34: w = int(W*0.5**i), h = int(H*0.5**i)."""
    return [(int(W * 0.5 ** i), int(H * 0.5 ** i)) for i in range(levels)]


def raster_level(xyz, M, W, H, threads=1):
    """One camera, one scale -> (index int32 [H,W], depth float32 [H,W])."""
    xyz, M = _f32(xyz), _f32(M).reshape(16)
    idx = np.empty((H, W), np.int32)
    dep = np.empty((H, W), np.float32)
    if threads > 1:
        _lib().oracle_raster_level_mt(_p(xyz, C.c_float), xyz.shape[0], _p(M, C.c_float), W, H,
                                      _p(idx, C.c_int32), _p(dep, C.c_float), threads)
    else:
        _lib().oracle_raster_level(_p(xyz, C.c_float), xyz.shape[0], _p(M, C.c_float), W, H,
                                   _p(idx, C.c_int32), _p(dep, C.c_float))
    return idx, dep


def raster_multiscale(xyz, M, W, H, levels=5, threads=1):
    """One camera, `levels` scales rasterised directly -> lists of (index, depth) per level."""
    xyz, M = _f32(xyz), _f32(M).reshape(16)
    sizes = level_sizes(W, H, levels)
    tot = sum(w * h for w, h in sizes)
    idx = np.empty(tot, np.int32)
    dep = np.empty(tot, np.float32)
    _lib().oracle_raster_multiscale(_p(xyz, C.c_float), xyz.shape[0], _p(M, C.c_float), W, H, levels,
                                    _p(idx, C.c_int32), _p(dep, C.c_float), threads)
    out_i, out_d, off = [], [], 0
    for w, h in sizes:
        out_i.append(idx[off:off + w * h].reshape(h, w))
        out_d.append(dep[off:off + w * h].reshape(h, w))
        off += w * h
    return out_i, out_d


def index_to_float(idx):
    idx = np.ascontiguousarray(idx, dtype=np.int32)
    out = np.empty(idx.shape, np.float32)
    _lib().oracle_index_to_float(_p(idx, C.c_int32), idx.size, _p(out, C.c_float))
    return out


def gather_chw(texture_cn, idx):
    """texture (C,N) channel-major, idx int32 [H,W] -> (C,H,W)  (texture.py:55-63)."""
    t = _f32(texture_cn)
    idx = np.ascontiguousarray(idx, dtype=np.int32)
    Cc, n = t.shape
    out = np.empty((Cc,) + idx.shape, np.float32)
    _lib().oracle_gather_chw(_p(t, C.c_float), n, Cc, _p(idx, C.c_int32), idx.size, _p(out, C.c_float))
    return out


def gather_backward_chw(grad_chw, idx, n):
    g = _f32(grad_chw)
    idx = np.ascontiguousarray(idx, dtype=np.int32)
    Cc = g.shape[0]
    out = np.zeros((Cc, n), np.float32)
    _lib().oracle_gather_backward_chw(_p(g, C.c_float), _p(idx, C.c_int32), idx.size, Cc, n,
                                      _p(out, C.c_float))
    return out
