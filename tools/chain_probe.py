"""Kernel boundary vs grid barrier (run on the GPU box; needs the debug library: python -m read_amd.build --debug).

    python tools/chain_probe.py
"""
import os
os.environ.setdefault("READ_HIP_DEBUG", "1")
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from read_amd import _lib  # noqa: E402

L = _lib.lib()
dev = torch.device("cuda", 0)
cf = torch.zeros(2, dtype=torch.int32, device=dev)


def run(mode, blocks, phases, fpb, iters=50):
    buf = torch.zeros(blocks * fpb, dtype=torch.float32, device=dev)
    for _ in range(3):
        _lib.check(L.read_debug_chain_probe(mode, blocks, phases, fpb, buf.data_ptr(), cf.data_ptr(), _lib.stream_ptr()))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        _lib.check(L.read_debug_chain_probe(mode, blocks, phases, fpb, buf.data_ptr(), cf.data_ptr(), _lib.stream_ptr()))
    e1.record()
    torch.cuda.synchronize()
    assert int(cf[1]) == 0, "a grid barrier timed out"
    return 1e3 * e0.elapsed_time(e1) / iters


for blocks in (256, 512, 1024):
    for fpb in (256, 4096):
        for phases in (1, 5):
            t0, t1 = run(0, blocks, phases, fpb), run(1, blocks, phases, fpb)
            print("blocks %4d  %6d B/block  phases %d:  launches %7.2f us   one launch + barriers %7.2f us" % (blocks, 4 * fpb, phases, t0, t1))
