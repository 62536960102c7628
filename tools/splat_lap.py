"""The rasteriser over ONE WHOLE LAP of the 256-pose sweep (slab or street scene): per block of 16 poses the per-kernel times
(read_splat_profile_last) and the chunk counters — the cost of a frame depends on the pose; a few dozen poses are not the sweep.

    python tools/splat_lap.py [slab|street] [key=value ...]
"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from read_amd import _lib, camera, synthetic                  # noqa: E402
from read_amd.raster import PointCloudRasterizer              # noqa: E402

scene = sys.argv[1] if len(sys.argv) > 1 else "slab"
N, H = (30_000_000, 352) if scene == "slab" else (10_000_000, 368)
W = 1216
L = _lib.lib()
for kv in sys.argv[2:]:
    k, v = kv.split("=")
    _lib.check(L.read_tuning_set(k.encode(), int(v)))
xyz = synthetic.make_cloud(N) if scene == "slab" else synthetic.make_street_cloud(N)
proj = synthetic.make_proj(W, H)
r = PointCloudRasterizer(xyz)
poses = [camera.total_matrix(proj, synthetic.sweep_pose(k)) for k in range(256)]
idx0, dep0 = r.render(poses[0], W, H)
call = r.bind(W, H, 5, (idx0, dep0), poses)
for k in range(256):                                          # one untimed lap: lists, seeds and marks settled
    call(k, (k + 1) % 256)
torch.cuda.synchronize()
# whole-lap time, events around the lap
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for k in range(256):
    call(k, (k + 1) % 256)
e1.record()
torch.cuda.synchronize()
print("%s %s: %.2f us/frame over the lap (announced, pre-bound calls)" % (scene, " ".join(sys.argv[2:]), 1e3 * e0.elapsed_time(e1) / 256), flush=True)
# per-kernel profile, pose by pose
_lib.check(L.read_tuning_set(b"splat_prof", 1))
ms = (C.c_float * 5)()
rows = []
for k in range(256):
    call(k, (k + 1) % 256)
    _lib.check(L.read_splat_profile_last(ms))
    rows.append(list(ms))
_lib.check(L.read_tuning_set(b"splat_prof", 0))
rows = np.asarray(rows) * 1e3
# counters, pose by pose
_lib.check(L.read_tuning_set(b"splat_stats", 1))
cnt = []
prev = r._ws[64:64 + 128].view(torch.int64).clone()
for k in range(256):
    call(k, (k + 1) % 256)
    torch.cuda.synchronize()
    cur = r._ws[64:64 + 128].view(torch.int64).clone()
    cnt.append((cur - prev).cpu().numpy())
    prev = cur
_lib.check(L.read_tuning_set(b"splat_stats", 0))
cnt = np.asarray(cnt, dtype=np.float64)
print("poses     pass_a  merge  pass_b  resolve+next |  A points   A items  A atomics  B culled  B run  B points")
for b in range(0, 256, 16):
    m, c = rows[b:b + 16].mean(0), cnt[b:b + 16].mean(0)
    print("%3d-%3d  %7.1f %6.1f %7.1f %9.1f     | %9.0f %9.0f %9.0f %9.0f %6.0f %9.0f" % (b, b + 15, m[1], m[2], m[3], m[4], c[0], c[8], c[2], c[9], c[11], c[4]))
m = rows.mean(0)
print("lap mean %7.1f %6.1f %7.1f %9.1f   (events around every launch: each figure carries ~2 us of event overhead)" % (m[1], m[2], m[3], m[4]))
