"""Synthetic scene / camera / weight recipes.

There is no network for kitti6 or checkpoints, so the benchmark and the parity
tests run on a seeded KITTI-like slab and seeded random UNet weights.  Everything
here is NumPy ``default_rng`` so the same bytes are produced on any box.
"""
import zlib

import numpy as np

from .camera import get_proj_matrix

DEFAULT_SEED = 2019          # the reference's --seed default (src/train.py:512)


def make_cloud(n_points, seed=DEFAULT_SEED):
    """xyz float32 (N,3): x~U(-60,60), y~U(-4,12), z~U(-120,-1) (GL camera looks down -z)."""
    rng = np.random.default_rng(seed)
    xyz = np.empty((n_points, 3), np.float32)
    xyz[:, 0] = rng.uniform(-60, 60, n_points)
    xyz[:, 1] = rng.uniform(-4, 12, n_points)
    xyz[:, 2] = rng.uniform(-120, -1, n_points)
    return xyz


def make_street_cloud(n_points, seed=DEFAULT_SEED):
    """Surface-like stand-in for a KITTI scene (BASELINE configs[1]; the real kitti6 scan cannot be downloaded here):
    a road plane, two facade walls with recessed openings, box-shaped vehicles and pole / foliage blobs along a 160 m
    street, seen from a camera 1.7 m above the road looking down -z.  Points lie ON surfaces (2 cm noise), densities
    differ by two orders of magnitude between parts and the order in memory is a random shuffle — unlike the uniform
    slab of make_cloud this has real occlusion (what is behind a wall is never visible) and empty sky."""
    rng = np.random.default_rng([seed, 77])
    n_ground, n_wall, n_car = int(0.35 * n_points), int(0.40 * n_points), int(0.10 * n_points)
    n_blob = n_points - n_ground - n_wall - n_car
    parts = []
    g = np.empty((n_ground, 3))
    g[:, 0] = rng.uniform(-25, 25, n_ground)
    g[:, 1] = -1.7 + 0.02 * rng.standard_normal(n_ground)
    g[:, 2] = rng.uniform(-160, 2, n_ground)
    parts.append(g)
    w = np.empty((n_wall, 3))
    side = rng.integers(0, 2, n_wall) * 2 - 1
    w[:, 2] = rng.uniform(-160, 2, n_wall)
    w[:, 1] = rng.uniform(-1.7, 11, n_wall)
    recess = ((np.floor(w[:, 2] / 7.0).astype(np.int64) % 3) == 0) * 2.5          # every third 7 m bay is set back
    w[:, 0] = side * (9.0 + recess) + 0.02 * rng.standard_normal(n_wall)
    parts.append(w)
    n_cars = 24
    centres = np.stack([rng.uniform(-6, 6, n_cars), np.full(n_cars, -0.95), rng.uniform(-150, -6, n_cars)], 1)
    half = np.array([0.9, 0.75, 2.1])
    c = rng.integers(0, n_cars, n_car)
    u = rng.uniform(-1, 1, (n_car, 3))
    face = rng.integers(0, 3, n_car)
    u[np.arange(n_car), face] = np.sign(u[np.arange(n_car), face])               # snap to one face of the box
    parts.append(centres[c] + u * half + 0.01 * rng.standard_normal((n_car, 3)))
    n_blobs = 60
    bc = np.stack([rng.choice([-1.0, 1.0], n_blobs) * rng.uniform(6.5, 8.5, n_blobs), rng.uniform(0, 6, n_blobs),
                   rng.uniform(-155, -3, n_blobs)], 1)
    b = rng.integers(0, n_blobs, n_blob)
    parts.append(bc[b] + rng.standard_normal((n_blob, 3)) * np.array([0.5, 1.2, 0.5]))
    xyz = np.concatenate(parts).astype(np.float32)
    return np.ascontiguousarray(xyz[rng.permutation(n_points)])


def make_descriptors(n_points, channels=8, seed=DEFAULT_SEED):
    """Descriptors float32 (C,N) ~ U(0,1) — PointTexture init_method='rand' (texture.py:25-26)."""
    rng = np.random.default_rng(seed + 1)
    return rng.random((channels, n_points), dtype=np.float32)


def make_intrinsics(W, H, f=720.0):
    return np.array([[f, 0, W / 2.0], [0, f, H / 2.0], [0, 0, 1.0]], np.float64)


def make_proj(W, H, f=720.0, znear=0.1, zfar=1000.0):
    """Dataset planes 0.1/1000 (READ/datasets/dynamic.py:111-112)."""
    return get_proj_matrix(make_intrinsics(W, H, f), (W, H), znear, zfar).astype(np.float32)


def sweep_pose(k):
    """Pose k of the 256-pose novel-view sweep: translate(0,0,-0.3k) o yaw(2deg*sin(2*pi*k/64)); cam->world."""
    a = np.deg2rad(2.0) * np.sin(2 * np.pi * k / 64.0)
    c, s = np.cos(a), np.sin(a)
    R = np.array([[c, 0, s, 0], [0, 1, 0, 0], [-s, 0, c, 0], [0, 0, 0, 1]], np.float64)
    T = np.eye(4)
    T[2, 3] = -0.3 * k
    return (T @ R).astype(np.float32)


# ----------------------------------------------------------------------------------------------
# UNet weights: per-tensor streams keyed by the state-dict name, so that any subset can be
# regenerated independently and the recipe does not depend on torch's RNG or module order.
# ----------------------------------------------------------------------------------------------
def _rng_for(name, seed):
    return np.random.default_rng([seed, zlib.crc32(name.encode())])


def make_unet_state(spec, seed=DEFAULT_SEED):
    """Random UNet state: dict name -> float32 ndarray, for every BasicConv in ``spec``
    (an iterable of (path, cin, cout, k)).  conv weights/biases ~ U(-b, b) with
    b = 1/sqrt(cin*k*k) (the Conv2d default-init bound); BN gamma~U(0.5,1.5), beta~N(0,0.1),
    running_mean~N(0,0.1), running_var~U(0.5,1.5) (non-trivial eval-mode statistics)."""
    st = {}
    for path, cin, cout, k in spec:
        bound = 1.0 / np.sqrt(cin * k * k)
        for conv in ("conv_f", "conv_m"):
            n = f"{path}.block.{conv}"
            st[n + ".weight"] = _rng_for(n + ".weight", seed).uniform(
                -bound, bound, (cout, cin, k, k)).astype(np.float32)
            st[n + ".bias"] = _rng_for(n + ".bias", seed).uniform(-bound, bound, cout).astype(np.float32)
        n = f"{path}.block.norm"
        st[n + ".weight"] = _rng_for(n + ".weight", seed).uniform(0.5, 1.5, cout).astype(np.float32)
        st[n + ".bias"] = (0.1 * _rng_for(n + ".bias", seed).standard_normal(cout)).astype(np.float32)
        st[n + ".running_mean"] = (0.1 * _rng_for(n + ".rm", seed).standard_normal(cout)).astype(np.float32)
        st[n + ".running_var"] = _rng_for(n + ".rv", seed).uniform(0.5, 1.5, cout).astype(np.float32)
        st[n + ".num_batches_tracked"] = np.zeros((), np.int64)
    return st
