"""Is the training step bound by the host or by the device?  (run on the GPU box: python tools/train_host_probe.py)

The UNet's training forward + backward + Adam at 8 crops of S x S for several S: host time per step WITHOUT a sync (what the
Python / launch path costs) next to the step time WITH one.  If the host time at S = 256 is far above the host time at a size
whose kernels are negligible, the difference is time spent blocked on the full launch queue, not Python."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from read_amd import synthetic  # noqa: E402
from read_amd.unet import UNet, weight_spec  # noqa: E402

state = synthetic.make_unet_state(weight_spec(), 3)
net = UNet()
net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in state.items()})
net.cuda().eval()
opt = torch.optim.Adam(net.parameters(), lr=1e-6, fused=True)
B = 8
for S in (32, 64, 128, 256):
    rng = np.random.default_rng(S)
    xs = [torch.from_numpy(rng.random((B, 8, S >> l, S >> l)).astype(np.float32)).cuda() for l in range(4)]
    g = torch.from_numpy(rng.standard_normal((B, 3, S, S)).astype(np.float32)).cuda() * 1e-3
    host, wall = [], []
    for it in range(9):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = net(*xs)
        t1 = time.perf_counter()
        out.backward(g)
        t2 = time.perf_counter()
        opt.step()
        opt.zero_grad()
        t3 = time.perf_counter()
        torch.cuda.synchronize()
        t4 = time.perf_counter()
        if it >= 3:
            host.append((t1 - t0, t2 - t1, t3 - t2))
            wall.append(t4 - t0)
    h = np.median(np.array(host), axis=0) * 1e3
    print({"crop": S, "host_forward_ms": round(float(h[0]), 2), "host_backward_ms": round(float(h[1]), 2),
           "host_adam_ms": round(float(h[2]), 2), "host_total_ms": round(float(h.sum()), 2),
           "step_with_sync_ms": round(float(np.median(wall)) * 1e3, 2)}, flush=True)
