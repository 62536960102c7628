"""ctypes front-end of oracle/_ref/libpcpr_ref.so — the reference's own DepthProject source
compiled for the CPU and run serially (ORACLE — test infrastructure only; see build_ref.sh)."""
import ctypes as C
import os

import numpy as np

PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref", "libpcpr_ref.so")


def available() -> bool:
    return os.path.exists(PATH)


def pcpr_ref_forward(xyz, total_m, W, H):
    """Same contract as the reference's pcpr.forward: -> (index f32 [B,H,W], depth f32 [B,H,W])."""
    L = C.CDLL(PATH)
    f32p = C.POINTER(C.c_float)
    L.pcpr_ref_forward.argtypes = [f32p, C.c_int64, f32p, C.c_int, C.c_int, C.c_int, f32p, f32p]
    L.pcpr_ref_forward.restype = None
    xyz = np.ascontiguousarray(xyz, np.float32)
    total_m = np.ascontiguousarray(total_m, np.float32).reshape(-1, 16)
    B = total_m.shape[0]
    idx = np.empty((B, H, W), np.float32)
    dep = np.empty((B, H, W), np.float32)
    L.pcpr_ref_forward(xyz.ctypes.data_as(f32p), xyz.shape[0], total_m.ctypes.data_as(f32p), B, W, H,
                       idx.ctypes.data_as(f32p), dep.ctypes.data_as(f32p))
    return idx, dep
