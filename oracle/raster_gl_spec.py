"""ORACLE — TEST INFRASTRUCTURE ONLY.

The GL twin's pixel coverage written from the OpenGL specification's TEXT, independently of raster.c's
``floor(u +- (s - 1) / 2)`` formulation, to pin the coverage rule for point sizes > 1 (odd AND even):

* OpenGL 4.6 core, section 13.7 (primitive clipping): "If the primitive under consideration is a point ..., then clipping
  passes it unchanged if it lies within the clip volume; otherwise, it is discarded."  -> a point is kept or dropped by its
  CENTRE; its square is never clipped against the frustum, only against the window.
* section 14.4 (points), GL_PROGRAM_POINT_SIZE enabled (READ/gl/render.py:55): "the derived point size is taken from the
  ... shader built-in gl_PointSize ... and clamped to the implementation-dependent point size range" (no rounding in the
  core profile; range taken as [1, 4096]).
* section 14.4.1 (basic point rasterization): "Point rasterization produces a fragment for each framebuffer pixel whose
  center lies inside a square centered at the point's (x_w, y_w), with side length equal to the current point size. ...
  all fragments produced in rasterizing a point are assigned the same associated data, which are those of the vertex".
* section 13.8.1 (viewport): x_w = (x_d + 1) p_x / 2 + o_x, y_w = (y_d + 1) p_y / 2 + o_y with window row 0 at the BOTTOM;
  the reference flips the rows afterwards (READ/datasets/dynamic.py:88-97).  Depth z_w = (z_d + 1) / 2 (default depth range).
* section 17.3.6 (depth test GL_LESS, READ/gl/render.py:56 default func): an incoming fragment replaces the stored one only
  if its depth is strictly smaller -> among equal depths the point drawn FIRST (smallest index) stays.

"Lies inside" is evaluated here as the half-open interval  x_w - s/2 <= c < x_w + s/2  (the top-left style fill convention
GPUs apply); a pixel centre closer than EPS to either edge is reported as ambiguous instead of being decided, so the
comparison with raster.c is exact everywhere else.  Pure Python / NumPy loops: small clouds only.

Vertex shader side (READ/gl/programs.py:121-128,183-192): gl_Position = P V M x (+ perturb on x, y);
point_size = global_point_size, or the a_point_size attribute when that is < 1; gl_PointSize = max(min_point_size,
point_size / gl_Position.z) for "ps" tokens, else point_size.
"""
import numpy as np

EPS = 1e-4


def raster_level_gl_spec(xyz, M, W, H, point_size=1.0, relative=False, min_point_size=1.0, point_sizes=None, discard=None,
                         perturb=None):
    """-> (index int32 [H,W], depth fp32 [H,W], ambiguous bool [H,W]); row 0 = image top (after the reference's flip)."""
    xyz = np.asarray(xyz, np.float32)
    M = np.asarray(M, np.float32).reshape(4, 4)
    n = xyz.shape[0]
    one = np.float32(1.0)
    x, y, z = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    with np.errstate(all="ignore"):
        c = [((M[k, 0] * x + M[k, 1] * y) + M[k, 2] * z) + M[k, 3] * one for k in range(4)]     # fp32, left to right
        if perturb is not None:
            pt = np.asarray(perturb, np.float32).reshape(n, 2)
            c[0] = c[0] + pt[:, 0]
            c[1] = c[1] + pt[:, 1]
        ndc = [c[k] / c[3] for k in range(3)]
    depth_buf = np.full((H, W), np.inf, np.float64)       # GL window rows: 0 = bottom
    index_buf = np.zeros((H, W), np.int32)
    depth_val = np.zeros((H, W), np.float32)
    ambiguous = np.zeros((H, W), bool)
    for i in range(n):
        if discard is not None and discard[i]:
            continue
        nx, ny, nz = (float(ndc[0][i]), float(ndc[1][i]), float(ndc[2][i]))
        if not (np.isfinite(nx) and np.isfinite(ny) and np.isfinite(nz)):
            continue
        if nx < -1 or nx > 1 or ny < -1 or ny > 1 or nz < -1 or nz > 1:        # clipped by its centre (13.7)
            continue
        xw = (nx + 1.0) * W / 2.0                                               # 13.8.1, float64 from the fp32 NDC
        yw = (ny + 1.0) * H / 2.0
        if not (0.0 <= xw < W and 0.0 < yw <= H):                               # centre on the window's last edge: no pixel of its own
            continue
        ps = 0.0 if point_sizes is not None else float(point_size)     # set_point_sizes zeroes the global size (programs.py:344)
        if ps < 1.0 and point_sizes is not None:
            ps = float(point_sizes[i])
        if relative:
            ps = max(float(min_point_size), float(np.float32(ps) / c[2][i]))
        s = min(max(ps, 1.0), 4096.0)
        zw = (np.float32(ndc[2][i]) + one) * np.float32(0.5)                    # fp32 like the depth buffer value
        lo_x, hi_x, lo_y, hi_y = xw - s / 2.0, xw + s / 2.0, yw - s / 2.0, yw + s / 2.0
        for j in range(max(int(np.floor(lo_y - 0.5)) - 1, 0), min(int(np.ceil(hi_y - 0.5)) + 1, H - 1) + 1):
            cy = j + 0.5
            near_y = min(abs(cy - lo_y), abs(cy - hi_y)) < EPS
            in_y = lo_y <= cy < hi_y
            if not (in_y or near_y):
                continue
            for k in range(max(int(np.floor(lo_x - 0.5)) - 1, 0), min(int(np.ceil(hi_x - 0.5)) + 1, W - 1) + 1):
                cx = k + 0.5
                near_x = min(abs(cx - lo_x), abs(cx - hi_x)) < EPS
                in_x = lo_x <= cx < hi_x
                if not (in_x or near_x):
                    continue
                if near_x or near_y:
                    ambiguous[j, k] = True
                    continue
                if float(zw) < depth_buf[j, k]:                                  # GL_LESS, points drawn in index order
                    depth_buf[j, k] = float(zw)
                    depth_val[j, k] = zw
                    index_buf[j, k] = i
    return index_buf[::-1].copy(), depth_val[::-1].copy(), ambiguous[::-1].copy()
