"""Per-kernel view of the rasteriser over consecutive sweep poses (run under rocprofv3 --kernel-trace --stats) and the
chunk / point counters of the cell-ordered passes (read_tuning_set("splat_stats", 1))."""
import sys, os
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from read_amd import _lib, camera, synthetic
from read_amd.raster import PointCloudRasterizer

N = int(sys.argv[1]) if len(sys.argv) > 1 else 30_000_000
W, H = 1216, int(os.environ.get('SPLAT_PROBE_H', '352'))
L = _lib.lib()
for kv in sys.argv[2:]:
    k, v = kv.split("=")
    _lib.check(L.read_tuning_set(k.encode(), int(v)))
xyz = synthetic.make_cloud(N, 2019) if os.environ.get("SPLAT_PROBE_SCENE", "slab") == "slab" else synthetic.make_street_cloud(N)
proj = synthetic.make_proj(W, H)
r = PointCloudRasterizer(xyz)
poses = [camera.total_matrix(proj, synthetic.sweep_pose(k)) for k in range(12)]
r.render(poses[0], W, H)
torch.cuda.synchronize()
ANN = os.environ.get("SPLAT_PROBE_ANNOUNCE", "1") == "1"
poses = [camera.total_matrix(proj, synthetic.sweep_pose(k)) for k in range(64)]
for rep in range(2):                                          # second repetition: clocks and sticky lists settled
    r.render(poses[0], W, H, next_total=poses[1] if ANN else None)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for k in range(1, 61):
        r.render(poses[k], W, H, next_total=poses[k + 1] if ANN else None)
    e1.record()
    torch.cuda.synchronize()
print("ms/frame %.4f (counters off, %s)" % (e0.elapsed_time(e1) / 60, "announced" if ANN else "unannounced"))
if os.environ.get("SPLAT_PROBE_STATS", "1") == "0":
    sys.exit(0)
_lib.check(L.read_tuning_set(b"splat_stats", 1))             # (the counters themselves cost ~0.3 ms per pass)
r.render(poses[0], W, H)
torch.cuda.synchronize()
hdr0 = r._ws[64:64 + 128].clone()
for k in range(1, 11):
    r.render(poses[k], W, H)
torch.cuda.synchronize()
st = (r._ws[64:64 + 128].view(torch.int64) - hdr0.view(torch.int64)).cpu().numpy() / 10.0
names = ["A points in strip", "A early-z reads", "A atomics", "-", "B points in strip", "B early-z reads", "B atomics", "-",
         "A items run", "B items culled", "-", "B items run"]
for n_, v in zip(names, st):
    print("  %-30s %12.0f" % (n_, v))
