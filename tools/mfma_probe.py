"""Sustained fp32 MFMA ceiling (v_mfma_f32_32x32x2_f32) at several occupancies and durations."""
import os
os.environ.setdefault("READ_HIP_DEBUG", "1")   # the probes live in libreadhip_debug.so only (python -m read_amd.build --debug)
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from read_amd import _lib  # noqa: E402

L = _lib.lib()
scratch = torch.zeros(256 * 8192, device="cuda")
res = []
for nacc in (4, -4):
    for wg_per_cu in (2,):
        for iters in (2000, 20000, 100000):
            blocks = 256 * wg_per_cu
            for _ in range(2):
                _lib.check(L.read_debug_mfma_probe(blocks, 200, nacc, scratch.data_ptr(), _lib.stream_ptr()))
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            _lib.check(L.read_debug_mfma_probe(blocks, iters, nacc, scratch.data_ptr(), _lib.stream_ptr()))
            e1.record()
            e1.synchronize()
            ms = e0.elapsed_time(e1)
            flops = blocks * 4 * iters * abs(nacc) * 4 * 4096.0
            row = {"nacc": nacc, "wg_per_cu": wg_per_cu, "iters": iters, "ms": ms, "tflops": flops / ms / 1e9}
            print(row, flush=True)
            res.append(row)
json.dump(res, open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/mfma_probe.json", "w"), indent=1)
