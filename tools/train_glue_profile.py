"""Which torch operations of a training step launch the glue kernels (fills, copies, adds), and from which source line?
(run on the GPU box: python tools/train_glue_profile.py)

The UNet's training forward + backward + Adam at 8 crops of 256 x 256 under torch.profiler with Python stacks: per aten operator
that launches a kernel outside libreadhip.so, the number of calls per step and the innermost frames of read_amd/ that issued them."""
import collections
import os
import sys

import numpy as np
import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from read_amd import synthetic  # noqa: E402
from read_amd.unet import UNet, weight_spec  # noqa: E402

state = synthetic.make_unet_state(weight_spec(), 3)
net = UNet()
net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in state.items()})
net.cuda().eval()
opt = torch.optim.Adam(net.parameters(), lr=1e-6, fused=True)
B, S = 8, 256
rng = np.random.default_rng(S)
xs = [torch.from_numpy(rng.random((B, 8, S >> l, S >> l)).astype(np.float32)).cuda() for l in range(4)]
g = torch.from_numpy(rng.standard_normal((B, 3, S, S)).astype(np.float32)).cuda() * 1e-3


def step():
    out = net(*xs)
    out.backward(g)
    opt.step()
    opt.zero_grad()


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
WATCH = ("aten::zeros", "aten::zero_", "aten::fill_", "aten::copy_", "aten::add", "aten::add_", "aten::mul", "aten::cat", "aten::sum",
         "aten::clone", "aten::contiguous", "aten::index_select", "aten::repeat_interleave", "aten::zeros_like", "aten::empty_like",
         "aten::slice_backward", "aten::select_backward", "aten::index_put_", "aten::_foreach_add_", "aten::_fused_adam_")
sites = collections.Counter()
totals = collections.Counter()
for ev in prof.events():
    if ev.name not in WATCH:
        continue
    totals[ev.name] += 1
    frames = [f for f in (ev.stack or []) if "read_amd/" in f or "tools/" in f]
    where = frames[0].split("read_amd/")[-1] if frames else ("autograd engine / optimizer" if ev.stack is not None else "?")
    sites[(ev.name, where)] += 1
print("aten operator calls in one step:", dict(totals))
for (name, where), n in sorted(sites.items(), key=lambda kv: -kv[1])[:60]:
    print("%5d  %-26s %s" % (n, name, where))
