"""Camera / clip-space conventions of READ's render path (host side, per frame: 16 floats).

``get_proj_matrix`` mirrors READ/gl/utils.py:123-150 (OpenGL-style projection from a
pinhole K; column-vector convention ``clip = P @ x_cam``; the camera looks down -z).
``total_matrix`` mirrors src/READ/gl/myrender.py:28-30 (``proj @ inv(view)``, fp32; the
result is an INPUT to the rasteriser — it is computed once on the host and the same 16
floats go to every consumer, never re-derived on the device).
"""
import numpy as np


def get_proj_matrix(K, image_size, znear=.01, zfar=1000.):
    """4x4 projection such that ``ndc = (P @ [x,y,z,1]) / w`` (READ/gl/utils.py:123-150)."""
    K = np.asarray(K)
    width, height = image_size
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    P = np.zeros((4, 4))
    P[0, 0] = 2.0 * fx / width
    P[1, 1] = 2.0 * fy / height
    P[0, 2] = 1.0 - 2.0 * cx / width
    P[1, 2] = 2.0 * cy / height - 1.0
    P[2, 2] = (zfar + znear) / (znear - zfar)
    P[2, 3] = 2.0 * zfar * znear / (znear - zfar)
    P[3, 2] = -1.0
    return P


def total_matrix(proj_matrix, view_matrix):
    """``proj @ inv(view)`` per batch item in the dtype of the inputs (myrender.py:28-30).

    ``view_matrix`` is camera->world.  Accepts (4,4) or (B,4,4); returns float32 (B,4,4)."""
    p = np.asarray(proj_matrix)
    v = np.asarray(view_matrix)
    if p.ndim == 2:
        p = p[None]
    if v.ndim == 2:
        v = v[None]
    return np.ascontiguousarray((p @ np.linalg.inv(v)).astype(np.float32))


def level_sizes(W, H, levels=5):
    """Per-scale raster sizes, ``int(W*0.5**i), int(H*0.5**i)`` (myrender.py:33-34)."""
    return [(int(W * 0.5 ** i), int(H * 0.5 ** i)) for i in range(levels)]
