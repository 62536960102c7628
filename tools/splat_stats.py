"""Counters of the MODE_HIZ rasteriser: how many points survive the LDS hi-z, how many atomics are issued."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from read_amd import _lib, camera, synthetic
from read_amd.raster import PointCloudRasterizer
W, H, N = 1216, 352, 30_000_000
xyz = synthetic.make_cloud(N)
proj = synthetic.make_proj(W, H)
r = PointCloudRasterizer(xyz)
L = _lib.lib()
L.read_tuning_set(b"splat_stats", 1)
for sub in (0, 8):
    L.read_tuning_set(b"splat_subset", sub)
    for mode in (7, 1):
        L.read_tuning_set(b"splat_mode", mode)
        for k in range(4):
            r.render(camera.total_matrix(proj, synthetic.sweep_pose(k)), W, H, 5)
            torch.cuda.synchronize()
            st = r._ws[64:64 + 64].view(torch.int64).cpu().numpy().copy()
            r._ws[64:128].zero_()
            print(f"subset {sub} mode {mode} frame {k}: A vis/surv/atomics {st[0]} {st[1]} {st[2]} | B vis/surv/atomics {st[4]} {st[5]} {st[6]}", flush=True)
        hiz = r._ws[256 + 8 * W * H * 8: 256 + 8 * W * H * 8 + 304 * 88 * 4].view(torch.float32)
        print("   finite hi-z blocks: %.3f, median bound %.5f" % (float(torch.isfinite(hiz).float().mean()), float(hiz[torch.isfinite(hiz)].median()) if torch.isfinite(hiz).any() else -1))
