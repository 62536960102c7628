"""What the rasteriser's frame-to-frame coherence is worth (run on the GPU box): the same 30 M-point slab rendered
(a) along the sweep (consecutive poses: what bench.py times), (b) jumping 97 poses of the 256-pose sweep per frame
("teleport": the warm start seeds and the sticky chunk lists of the previous frame are wrong), (c) with the warm start
switched off, (d) the very first frame of a fresh rasteriser.  Every frame of (b) is compared with the frame the coherent
walk produces for the same pose: the results must be identical, only the time may differ.

    python tools/splat_coherence.py [--points 30000000] [--out gpurun_out/splat_coherence.json]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from read_amd import _lib, camera, synthetic          # noqa: E402
from read_amd.raster import PointCloudRasterizer      # noqa: E402


def timed(r, Ms, order, W, H):
    e = [torch.cuda.Event(enable_timing=True) for _ in range(len(order) + 1)]
    e[0].record()
    outs = []
    for i, k in enumerate(order):
        idx, dep = r.render(Ms[k], W, H, 5)
        outs.append((idx[0].clone(), dep[0].clone()))
        e[i + 1].record()
    torch.cuda.synchronize()
    # the clones are inside the timed region of every variant alike (6.8 MB each: ~3 us)
    return [e[i].elapsed_time(e[i + 1]) for i in range(len(order))], outs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=30_000_000)
    ap.add_argument("--out", default="gpurun_out/splat_coherence.json")
    a = ap.parse_args()
    W, H = 1216, 352
    xyz = synthetic.make_cloud(a.points)
    proj = synthetic.make_proj(W, H)
    Ms = [camera.total_matrix(proj, synthetic.sweep_pose(k)) for k in range(256)]
    L = _lib.lib()
    r = PointCloudRasterizer(xyz)
    torch.cuda.synchronize()
    first, _ = timed(r, Ms, [0], W, H)                                   # (d) the very first frame
    walk = list(range(1, 65))
    t_walk, o_walk = timed(r, Ms, walk, W, H)                            # (a)
    jump = [(1 + 97 * i) % 256 for i in range(64)]
    t_jump, o_jump = timed(r, Ms, jump, W, H)                            # (b)
    ref = {k: o for k, o in zip(walk, o_walk)}
    same = [torch.equal(o[0], ref[k][0]) and torch.equal(o[1], ref[k][1]) for k, o in zip(jump, o_jump) if k in ref]
    _lib.check(L.read_tuning_set(b"splat_seeds", 0))
    t_cold, o_cold = timed(r, Ms, walk, W, H)                            # (c)
    _lib.check(L.read_tuning_set(b"splat_seeds", 1))
    same_cold = all(torch.equal(x[0], y[0]) and torch.equal(x[1], y[1]) for x, y in zip(o_cold, o_walk))
    med = lambda v: sorted(v)[len(v) // 2]
    bytes_algo = 12.0 * a.points + 8.0 * sum(w * h for (w, h) in camera.level_sizes(W, H, 5))
    res = {"points": a.points, "first_frame_ms": first[0],
           "coherent_sweep_ms_median": med(t_walk), "coherent_sweep_ms_max": max(t_walk[4:]),
           "teleport_ms_median": med(t_jump), "teleport_ms_max": max(t_jump),
           "no_warm_start_ms_median": med(t_cold),
           "teleport_frames_compared": len(same), "teleport_frames_identical": all(same), "no_warm_start_identical": same_cold,
           "frac_hbm": {k: bytes_algo / (v * 1e-3) / 8e12 for k, v in (("coherent", med(t_walk)), ("teleport", med(t_jump)),
                                                                       ("no_warm_start", med(t_cold)))}}
    print(json.dumps(res, indent=1))
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    json.dump(res, open(a.out, "w"), indent=1)
    if not (all(same) and same_cold):
        sys.exit(3)


if __name__ == "__main__":
    main()
