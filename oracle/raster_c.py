"""ctypes front-end of oracle/raster.c (ORACLE — test infrastructure only)."""
import ctypes as C
import os

import numpy as np

from . import build as _build

_LIB = None


class GLOpts(C.Structure):
    _fields_ = [("point_size", C.c_float), ("relative", C.c_int), ("min_point_size", C.c_float),
                ("discard", C.c_void_p), ("drop_threshold", C.c_uint32), ("drop_seed", C.c_uint32),
                ("perturb", C.c_void_p), ("perturb_amp", C.c_float), ("perturb_seed", C.c_uint32),
                ("point_sizes", C.c_void_p)]


def lib_path() -> str:
    return _build.OUT


def _lib():
    global _LIB
    if _LIB is None:
        _build.build()                     # no-op when liboracle_raster.so is newer than raster.c
        L = C.CDLL(_build.OUT)
        f32p, i32p = C.POINTER(C.c_float), C.POINTER(C.c_int32)
        L.oracle_raster_level.argtypes = [f32p, C.c_int64, f32p, C.c_int, C.c_int, i32p, f32p]
        L.oracle_raster_level_mt.argtypes = [f32p, C.c_int64, f32p, C.c_int, C.c_int, i32p, f32p, C.c_int]
        L.oracle_raster_multiscale.argtypes = [f32p, C.c_int64, f32p, C.c_int, C.c_int, C.c_int,
                                               i32p, f32p, C.c_int]
        L.oracle_project_points.argtypes = [f32p, C.c_int64, f32p, C.c_int, C.c_int, i32p, f32p, C.c_int]
        L.oracle_project_points.restype = None
        L.oracle_index_to_float.argtypes = [i32p, C.c_size_t, f32p]
        L.oracle_gather_chw.argtypes = [f32p, C.c_int64, C.c_int, i32p, C.c_size_t, f32p]
        L.oracle_gather_backward_chw.argtypes = [f32p, i32p, C.c_size_t, C.c_int, C.c_int64, f32p]
        L.oracle_raster_level_gl.argtypes = [f32p, C.c_int64, f32p, C.c_int, C.c_int, C.POINTER(GLOpts), i32p, f32p]
        L.oracle_drop_mask.argtypes = [C.c_int64, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint8)]
        L.oracle_perturb_array.argtypes = [C.c_int64, C.c_float, C.c_uint32, f32p]
        for fn in (L.oracle_raster_level_gl, L.oracle_drop_mask, L.oracle_perturb_array):
            fn.restype = None
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


def project_points(xyz, M, W, H, threads=1):
    """Projection alone (point_render.cu:110-147): per point the pixel index yy * W + xx (or -1: rejected) and the depth
    (0 for rejected points)."""
    xyz, M = _f32(xyz), _f32(M).reshape(16)
    pix = np.empty(xyz.shape[0], np.int32)
    dep = np.empty(xyz.shape[0], np.float32)
    _lib().oracle_project_points(_p(xyz, C.c_float), xyz.shape[0], _p(M, C.c_float), W, H, _p(pix, C.c_int32),
                                 _p(dep, C.c_float), threads)
    return pix, dep


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


def drop_threshold(p):
    """Probability -> the u32 threshold of the seeded drop (point i dropped iff rnd(i, seed, 0) < threshold)."""
    return int(min(max(float(p), 0.0), 1.0) * 4294967295.0)


def raster_level_gl(xyz, M, W, H, point_size=1.0, relative=False, min_point_size=1.0, discard=None, drop=None,
                    perturb=None, perturb_hash=None, point_sizes=None):
    """GL-twin rasterisation of ONE level at its own size (see raster.c): point sizes / "ps" splats, discard mask or
    seeded drop=(p, seed), perturb array (N,2) or seeded perturb_hash=(amp, seed) -> (index int32 [H,W], depth [H,W])."""
    xyz, M = _f32(xyz), _f32(M).reshape(16)
    o = GLOpts(float(point_size), int(bool(relative)), float(min_point_size), None, 0, 0, None, 0.0, 0, None)
    keep = []
    if point_sizes is not None:                    # NNScene.set_point_sizes: the global size becomes 0, the attribute rules
        psz = _f32(point_sizes).reshape(-1)
        keep.append(psz)
        o.point_sizes = psz.ctypes.data
        o.point_size = 0.0
    if discard is not None:
        dm = np.ascontiguousarray(discard, dtype=np.uint8)
        keep.append(dm)
        o.discard = dm.ctypes.data
    if drop is not None:
        o.drop_threshold, o.drop_seed = drop_threshold(drop[0]), int(drop[1]) & 0xffffffff
    if perturb is not None:
        pa = _f32(perturb)
        keep.append(pa)
        o.perturb = pa.ctypes.data
    if perturb_hash is not None:
        o.perturb_amp, o.perturb_seed = float(perturb_hash[0]), int(perturb_hash[1]) & 0xffffffff
    idx = np.empty((H, W), np.int32)
    dep = np.empty((H, W), np.float32)
    _lib().oracle_raster_level_gl(_p(xyz, C.c_float), xyz.shape[0], _p(M, C.c_float), W, H, C.byref(o),
                                  _p(idx, C.c_int32), _p(dep, C.c_float))
    return idx, dep


def drop_mask(n, p, seed):
    m = np.empty(n, np.uint8)
    _lib().oracle_drop_mask(n, drop_threshold(p), int(seed) & 0xffffffff, _p(m, C.c_uint8))
    return m.astype(bool)


def perturb_array(n, amp, seed):
    out = np.empty((n, 2), np.float32)
    _lib().oracle_perturb_array(n, float(amp), int(seed) & 0xffffffff, _p(out, C.c_float))
    return out
