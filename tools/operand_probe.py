"""fp32 MFMA rate by operand register pattern (run on the GPU box): python tools/operand_probe.py"""
import os
os.environ.setdefault("READ_HIP_DEBUG", "1")   # the probes live in libreadhip_debug.so only (python -m read_amd.build --debug)
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from read_amd import _lib  # noqa: E402

L = _lib.lib()
scratch = torch.zeros(256 * 8192, device="cuda")
cycles = torch.zeros(4096, dtype=torch.int64, device="cuda")
iters = 4000
for wps in (1, 2):
    for mode in (0, 1, 2, 3):
        for rep in range(2):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            _lib.check(L.read_debug_operand_probe(mode, 256 * wps, iters, scratch.data_ptr(), cycles.data_ptr(), _lib.stream_ptr()))
            e1.record()
            e1.synchronize()
        ms = e0.elapsed_time(e1)
        per = float(cycles[:256 * wps * 4].double().mean()) / (iters * 16)
        print({"waves_per_simd": wps, "mode": mode, "ticks_per_mfma_per_simd": per / wps, "ns_per_mfma_per_simd": 1e6 * ms / (iters * 16 * wps)}, flush=True)
