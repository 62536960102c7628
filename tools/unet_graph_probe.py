"""The UNet plan (105 launches, side streams for the SCM chains) replayed from a HIP graph against the eager enqueue, one frame at a
time on one stream (run on the GPU box): ms per frame, and the two outputs compared."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from read_amd import synthetic                                            # noqa: E402
from read_amd.unet import LAYOUT_LEAN, UNetEngine, pack_state, weight_spec   # noqa: E402

H, W = 352, 1216
dev = torch.device("cuda:0")
state = synthetic.make_unet_state(weight_spec())
eng = UNetEngine(torch.from_numpy(pack_state(state, layout=LAYOUT_LEAN)).to(dev), H, W)
x = [torch.randn(H >> l, W >> l, 8, device=dev) for l in range(4)]
out = torch.empty(H, W, 4, device=dev)


def timed(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / n, (time.perf_counter() - t0) * 1e3 / n


eager = lambda: eng.forward(*x, out=out, channels=4)     # noqa: E731
print("eager   : %.3f ms per frame (events), %.3f ms wall" % timed(eager))
ref = out.clone()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(3):
        eager()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        eager()
torch.cuda.synchronize()
out.zero_()
g.replay()
torch.cuda.synchronize()
print("graph output equals eager output:", bool(torch.equal(out, ref)))
print("graph   : %.3f ms per frame (events), %.3f ms wall" % timed(g.replay))
print("eager   : %.3f ms per frame (events), %.3f ms wall" % timed(eager))
