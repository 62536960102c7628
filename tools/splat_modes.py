"""A/B the rasteriser's knobs on the GPU box (read_tuning_set) over consecutive poses of the sweep, like bench.py.

    python tools/splat_modes.py [--points 30000000] [--scene slab|street] [--out gpurun_out/splat_modes.json]

Every variant must produce frames identical to the first one (all knobs of the release library are exact)."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from read_amd import _lib, camera, synthetic          # noqa: E402
from read_amd.raster import PointCloudRasterizer      # noqa: E402

DEFAULTS = {"splat_mode": 7, "splat_cells": 1, "splat_seeds": 1, "splat_near": 12, "splat_cells_sub": 0, "splat_items": 4,
            "splat_subset": 8, "splat_strips": 1, "splat_zl2": 0, "splat_wgs": 4, "splat_lds": 1, "splat_kslot": 0}
VARIANTS = [
    ("default: striped cell-ordered passes, zimg early-z, warm start", {}),
    ("no LDS table in front of the atomics", {"splat_lds": 0}),
    ("8 workgroups per CU", {"splat_wgs": 8}),
    ("scattered key image", {"splat_kslot": 2}),
    ("early-z loads from L2 (sc1)", {"splat_zl2": 1}),
    ("4 strips", {"splat_strips": 4}),
    ("2 strips", {"splat_strips": 2}),
    ("8 strips (XCD affinity of the bound image)", {"splat_strips": 8}),
    ("items per chunk 2", {"splat_items": 2}),
    ("items per chunk 1", {"splat_items": 1}),
    ("near split 6 points/pixel", {"splat_near": 6}),
    ("near split 24 points/pixel", {"splat_near": 24}),
    ("near split 48 points/pixel", {"splat_near": 48}),
    ("every 16th chunk in pass A", {"splat_cells_sub": 16}),
    ("no every-n-th chunk in pass A", {"splat_cells_sub": 0}),
    ("no warm start", {"splat_seeds": 0}),
    ("plain path: warm start + LDS hi-z over the unsorted cloud", {"splat_cells": 0}),
    ("plain path: agent atomics + early-z only", {"splat_cells": 0, "splat_mode": 1}),
    ("default again", {}),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=30_000_000)
    ap.add_argument("--scene", default="slab")
    ap.add_argument("--height", type=int, default=352)
    ap.add_argument("--only", type=int, default=-1)
    ap.add_argument("--out", default="gpurun_out/splat_modes.json")
    a = ap.parse_args()
    W, H = 1216, a.height
    xyz = synthetic.make_cloud(a.points) if a.scene == "slab" else synthetic.make_street_cloud(a.points)
    proj = synthetic.make_proj(W, H)
    Ms = [camera.total_matrix(proj, synthetic.sweep_pose(k)) for k in range(40)]     # a moving camera, like bench.py
    r = PointCloudRasterizer(xyz)
    L = _lib.lib()
    ref = None
    res = []
    bytes_algo = 12.0 * a.points + 8.0 * sum(w * h for (w, h) in camera.level_sizes(W, H, 5))
    for vi, (name, knobs) in enumerate(VARIANTS):
        if a.only >= 0 and vi != a.only:
            continue
        for k, v in {**DEFAULTS, **knobs}.items():
            _lib.check(L.read_tuning_set(k.encode(), v))
        for k in range(4):
            idx, dep = r.render(Ms[k], W, H, 5)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for k in range(32):
            idx, dep = r.render(Ms[4 + k], W, H, 5)
        e1.record()
        e1.synchronize()
        ms = e0.elapsed_time(e1) / 32
        cur = [i.clone() for i in idx] + [d.clone() for d in dep]
        if ref is None:
            ref = cur
        same = all(torch.equal(x, y) for x, y in zip(ref, cur))
        row = {"variant": name, "knobs": knobs, "ms": ms, "GBps": bytes_algo / ms / 1e6,
               "frac_hbm_8TBs": bytes_algo / ms / 1e6 / 8000.0, "identical_to_first": same}
        print(row, flush=True)
        res.append(row)
    for k, v in DEFAULTS.items():
        _lib.check(L.read_tuning_set(k.encode(), v))
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
