"""ORACLE — TEST INFRASTRUCTURE ONLY.

Independent NumPy restatement of READ's z-buffer projector, written without looking
at raster.c's control flow: vectorised fp32 arithmetic + a lexicographic sort to pick,
per pixel, (min depth, then min index).  Exists to cross-check raster.c.

Follows MyRender/CloudProjection/point_render.cu:110-121,135-159 and
helper_math.h:1252-1255 (dot evaluated left to right in fp32, no FMA — NumPy never fuses).
"""
import numpy as np


def project_np(xyz, M, W, H):
    """-> (accepted mask, pixel index, depth) all per point, fp32 arithmetic."""
    xyz = np.asarray(xyz, np.float32)
    M = np.asarray(M, np.float32).reshape(4, 4)
    x, y, z = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    one = np.float32(1.0)
    with np.errstate(all="ignore"):
        c = [((M[k, 0] * x + M[k, 1] * y) + M[k, 2] * z) + M[k, 3] * one for k in range(4)]
        nx, ny, nz = c[0] / c[3], c[1] / c[3], c[2] / c[3]
        ok = np.isfinite(nx) & np.isfinite(ny) & np.isfinite(nz)
        ok &= ~((nx < -1) | (nx > 1) | (ny < -1) | (ny > 1) | (nz < -1) | (nz > 1))
        u = (np.float32(W) * (nx + one)) * np.float32(0.5)
        v = (np.float32(H) * (one - ny)) * np.float32(0.5)
        d = (nz + one) * np.float32(0.5)
        u = np.where(ok, u, 0).astype(np.float32)
        v = np.where(ok, v, 0).astype(np.float32)
        xx = u.astype(np.int64)          # truncation toward zero
        yy = v.astype(np.int64)
    ok &= (xx >= 0) & (xx < W) & (yy >= 0) & (yy < H)
    return ok, yy * W + xx, d.astype(np.float32)


def raster_level_np(xyz, M, W, H):
    ok, pix, d = project_np(xyz, M, W, H)
    ids = np.nonzero(ok)[0]
    pix, d = pix[ok], d[ok]
    # depth >= 0 here, so its uint32 bit pattern orders like the float
    key = (d.view(np.uint32).astype(np.uint64) << np.uint64(32)) | ids.astype(np.uint64)
    order = np.lexsort((key, pix))
    pix_s, key_s = pix[order], key[order]
    first = np.ones(pix_s.shape[0], bool)
    first[1:] = pix_s[1:] != pix_s[:-1]
    idx = np.zeros(W * H, np.int32)
    dep = np.zeros(W * H, np.float32)
    win = key_s[first]
    idx[pix_s[first]] = (win & np.uint64(0xFFFFFFFF)).astype(np.int64).astype(np.int32)
    dep[pix_s[first]] = (win >> np.uint64(32)).astype(np.uint32).view(np.float32)
    return idx.reshape(H, W), dep.reshape(H, W)
