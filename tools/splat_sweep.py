"""Knob sweep of the cell-path rasteriser in ONE process (the cloud is built once): for every knob set, ms per frame over 60
consecutive sweep poses (announced next camera) and the chunk counters.  Run on the GPU box.

    python tools/splat_sweep.py [slab|street] "splat_near=12,splat_sticky=8" "splat_near=8,splat_sticky=1" ...
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from read_amd import _lib, camera, synthetic                  # noqa: E402
from read_amd.raster import PointCloudRasterizer              # noqa: E402

scene = sys.argv[1]
N, H = (30_000_000, 352) if scene == "slab" else (10_000_000, 368)
W = 1216
L = _lib.lib()
xyz = synthetic.make_cloud(N, 2019) if scene == "slab" else synthetic.make_street_cloud(N)
proj = synthetic.make_proj(W, H)
r = PointCloudRasterizer(xyz)
poses = [camera.total_matrix(proj, synthetic.sweep_pose(k)) for k in range(64)]
names = ["A points", "-", "A atomics", "-", "B points", "-", "B atomics", "-", "A items", "B culled", "-", "B run"]
defaults = {}
for ks in sys.argv[2:]:
    sets = [kv.split("=") for kv in ks.split(",") if kv]
    for k, v in sets:
        if k not in defaults:
            cur = torch.zeros(1, dtype=torch.int32)
            import ctypes as C
            c = C.c_int()
            _lib.check(L.read_tuning_get(k.encode(), C.byref(c)))
            defaults[k] = c.value
        _lib.check(L.read_tuning_set(k.encode(), int(v)))
    best = 1e9
    idx0, dep0 = r.render(poses[0], W, H)
    call = r.bind(W, H, 5, (idx0, dep0), poses)               # pre-bound arguments: the loop below is the device's, not the host's
    for rep in range(3):
        call(0, 1)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for k in range(1, 61):
            call(k, k + 1)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 60)
    pol = []
    for k in range(1, 13):                                     # the list-A policy evidence (SplatHeader::pol_near, pol_sticky), frame by frame
        call(k, k + 1)
        torch.cuda.synchronize()
        pol.append(tuple(int(v) for v in r._ws[16:24].view(torch.int32).cpu()))
    print("   policy evidence (near, sticky) per frame:", pol, flush=True)
    _lib.check(L.read_tuning_set(b"splat_stats", 1))
    r.render(poses[0], W, H)
    torch.cuda.synchronize()
    h0 = r._ws[64:64 + 128].clone()
    for k in range(1, 11):
        r.render(poses[k], W, H)
    torch.cuda.synchronize()
    st = (r._ws[64:64 + 128].view(torch.int64) - h0.view(torch.int64)).cpu().numpy() / 10.0
    _lib.check(L.read_tuning_set(b"splat_stats", 0))
    print("%-6s %-44s %.4f ms/frame  " % (scene, ks, best) + "  ".join("%s %.0f" % (n_, v) for n_, v in zip(names, st) if n_ != "-"), flush=True)
    for k, v in defaults.items():
        _lib.check(L.read_tuning_set(k.encode(), v))
