"""A/B the rasteriser's key-image policies on the GPU box (read_tuning_set("splat_mode", m)).

    python tools/splat_modes.py [--points 30000000] [--out gpurun_out/splat_modes.json]
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from read_amd import _lib, camera, synthetic          # noqa: E402
from read_amd.raster import PointCloudRasterizer      # noqa: E402

NAMES = {7: "warm start + LDS hi-z + agent atomics (default)", 0: "per-XCD images, workgroup-scope atomics", 1: "one image, agent atomics, sc1 early-z",
         3: "one image, agent atomics, system-scope early-z", 2: "probe: projection only",
         4: "probe: projection + sc1 early-z reads, no atomics", 5: "probe: projection + plain (L1) early-z reads",
         6: "probe: projection + atomics, no early-z"}
INVALID = (2, 4, 5, 6)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=30_000_000)
    ap.add_argument("--out", default="gpurun_out/splat_modes.json")
    a = ap.parse_args()
    W, H = 1216, 352
    xyz = synthetic.make_cloud(a.points)
    proj = synthetic.make_proj(W, H)
    Ms = [camera.total_matrix(proj, synthetic.sweep_pose(k)) for k in range(32)]     # a moving camera, like bench.py
    r = PointCloudRasterizer(xyz)
    L = _lib.lib()
    ref = None
    res = []
    bytes_algo = 12.0 * a.points + 8.0 * sum(w * h for (w, h) in camera.level_sizes(W, H, 5))
    for mode in (7, 1, 7004, 7008, 7016, 7032, 7):
        sub = 8
        pipe = 0
        if mode == 1001:            # mode 1 with the straightforward (non-pipelined) loop
            mode, pipe = 1, 0
        _lib.check(L.read_tuning_set(b"splat_pipe", pipe))
        if mode >= 7000:            # 70xx = mode 7 with bootstrap subset xx (0 = seeds only)
            sub, mode = mode - 7000, 7
            sub = 0 if sub == 7 else sub
        _lib.check(L.read_tuning_set(b"splat_subset", sub))
        _lib.check(L.read_tuning_set(b"splat_mode", mode))
        for k in range(3):
            idx, dep = r.render(Ms[k], W, H, 5)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for k in range(10):
            idx, dep = r.render(Ms[3 + k], W, H, 5)
        e1.record()
        e1.synchronize()
        ms = e0.elapsed_time(e1) / 10
        same = None
        if mode not in INVALID:
            cur = [i.clone() for i in idx] + [d.clone() for d in dep]
            if ref is None:
                ref = cur
            same = all(torch.equal(x, y) for x, y in zip(ref, cur))
        row = {"mode": mode, "subset": sub, "pipe": pipe, "name": NAMES[mode], "ms": ms, "GBps": bytes_algo / ms / 1e6,
               "frac_hbm_8TBs": bytes_algo / ms / 1e6 / 8000.0, "identical_to_first": same}
        print(row, flush=True)
        res.append(row)
        # the workspace may hold garbage after the projection-only mode: re-initialise
        _lib.check(L.read_splat_workspace_init(r._ws.data_ptr(), r._ws.numel(), _lib.stream_ptr()))
    _lib.check(L.read_tuning_set(b"splat_mode", 7))
    _lib.check(L.read_tuning_set(b"splat_subset", 8))
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
