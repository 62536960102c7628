"""Which Python lines of a training step launch the small torch kernels (fills, copies, elementwise): torch.profiler with stacks
around ONE step of bench.py's training workload (run on the GPU box).

    python tools/train_launch_census.py
"""
import collections
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from read_amd import train as T  # noqa: E402

sys.argv = ["bench.py", "--config", "train", "--no-cpu-baseline", "--steps", "3"]
a = bench.parse()
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
T.GRAPH_TRAIN = False


def census(step):
    from torch.profiler import ProfilerActivity, profile
    for i in range(3):
        step(i)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        step(3)
        torch.cuda.synchronize()
    by_line = collections.Counter()
    by_op = collections.Counter()
    for ev in prof.events():
        if ev.device_type is not None and str(ev.device_type).endswith("CUDA"):
            continue
        name = ev.name
        if not name.startswith("aten::") or name in ("aten::empty", "aten::empty_like", "aten::view", "aten::as_strided", "aten::detach",
                                                     "aten::reshape", "aten::permute", "aten::slice", "aten::select", "aten::unbind",
                                                     "aten::empty_strided", "aten::_unsafe_view", "aten::alias", "aten::expand", "aten::narrow",
                                                     "aten::transpose", "aten::unsqueeze", "aten::squeeze", "aten::result_type", "aten::to",
                                                     "aten::contiguous", "aten::clone", "aten::zeros", "aten::zeros_like", "aten::ones_like",
                                                     "aten::lift_fresh", "aten::t", "aten::item", "aten::_local_scalar_dense"):
            continue
        frame = next((f for f in (ev.stack or []) if "/read_amd/" in f or "bench.py" in f or "/torch/optim/" in f or "autograd" in f), "?")
        by_line[(name, frame)] += 1
        by_op[name] += 1
    print("ops per step:", sum(by_op.values()))
    for (name, frame), n in by_line.most_common(45):
        print("%5d  %-22s %s" % (n, name, frame[-110:]))


# run_train builds everything and calls step() in its loops: intercept the first timed loop through a tiny monkeypatch of range use
src = open(os.path.join(ROOT, "bench.py")).read()
marker = "    steps = steps if steps is not None else (a.steps if a.steps != 256 else 10)\n"
assert marker in src
src = src.replace(marker, "    import builtins\n    builtins._census_hook(step)\n    raise SystemExit(0)\n" + marker, 1)
import builtins  # noqa: E402
builtins._census_hook = census
g = {"__name__": "bench_census", "__file__": os.path.join(ROOT, "bench.py")}
exec(compile(src, os.path.join(ROOT, "bench.py"), "exec"), g)
try:
    g["run_train"](a, dev, steps=1, warm=1, cpu_timing=False, do_verify=False)
except SystemExit:
    pass
