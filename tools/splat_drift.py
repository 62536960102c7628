"""Does the rasteriser stage's time drift with how long it has been running (clock management)?  One process, the slab scene:
blocks of 256 frames (one lap of the sweep, next camera announced, pre-bound call), back to back for ~2.5 s, each block timed with
its own event pair; then 3 s of host sleep (device idle) and the same again; then 1 s of dense MFMA work (a torch fp32 matmul loop)
directly followed by rasteriser blocks.  Run on the GPU box.

    python tools/splat_drift.py
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from read_amd import camera, synthetic                        # noqa: E402
from read_amd.raster import PointCloudRasterizer              # noqa: E402

W, H, N = 1216, 352, 30_000_000
xyz = synthetic.make_cloud(N)
proj = synthetic.make_proj(W, H)
r = PointCloudRasterizer(xyz)
poses = [camera.total_matrix(proj, synthetic.sweep_pose(k)) for k in range(256)]
idx0, dep0 = r.render(poses[0], W, H)
call = r.bind(W, H, 5, (idx0, dep0), poses)


def blocks(n, tag):
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    evs[0].record()
    for b in range(n):
        for k in range(256):
            call(k, (k + 1) % 256)
        evs[b + 1].record()
    torch.cuda.synchronize()
    print(tag, " ".join("%.1f" % (1e3 * evs[b].elapsed_time(evs[b + 1]) / 256) for b in range(n)), "us/frame per block of 256", flush=True)


blocks(100, "cold start      :")
time.sleep(3.0)
blocks(40, "after 3 s idle  :")
a = torch.randn(8192, 8192, device="cuda")
t0 = time.time()
while time.time() - t0 < 1.0:
    b = a @ a
torch.cuda.synchronize()
blocks(40, "after 1 s matmul:")
