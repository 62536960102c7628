import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from read_amd import _lib, camera, synthetic
from read_amd.raster import PointCloudRasterizer
W, H, N = 1216, 352, 30_000_000
xyz = synthetic.make_cloud(N); proj = synthetic.make_proj(W, H)
r = PointCloudRasterizer(xyz); L = _lib.lib()
for sub in (0, 8, 16):
    L.read_tuning_set(b"splat_subset", sub)
    for k in range(6):
        r.render(camera.total_matrix(proj, synthetic.sweep_pose(k)), W, H, 5)
    torch.cuda.synchronize()
