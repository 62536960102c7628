"""The training step's rasteriser call in isolation: 8 cameras x 5 scales at 256x256 over the 10 M-point street scene
(MyRender.render's batch), as B cell-path frames (default) and on the plain pass (splat_cells_batch=0).  Run on the GPU box."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from read_amd import _lib, camera, synthetic                  # noqa: E402
from read_amd.raster import PointCloudRasterizer              # noqa: E402

L = _lib.lib()
for kv in sys.argv[1:]:
    k_, v_ = kv.split("=")
    _lib.check(L.read_tuning_set(k_.encode(), int(v_)))
S, B, N = 256, 8, 10_000_000
xyz = synthetic.make_street_cloud(N)
proj = synthetic.make_proj(S, S)
r = PointCloudRasterizer(xyz)
rng = np.random.default_rng(2019)
batches = [camera.total_matrix(proj, np.stack([synthetic.sweep_pose(int(k)) for k in rng.integers(0, 256, B)])) for _ in range(12)]
for knob in (1, 0):
    _lib.check(L.read_tuning_set(b"splat_cells_batch", knob))
    for M in batches[:2]:
        r.render(M, S, S, 5)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for M in batches[2:]:
        out = r.render(M, S, S, 5)
    e1.record()
    torch.cuda.synchronize()
    print("splat_cells_batch=%d: %.1f us per batch of %d cameras (256x256, 5 scales, %d points)" % (knob, 1e3 * e0.elapsed_time(e1) / 10, B, N), flush=True)
_lib.check(L.read_tuning_set(b"splat_cells_batch", 1))
