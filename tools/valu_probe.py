"""Cost of one fp32 vector instruction per wave, by form (run on the GPU box): python tools/valu_probe.py"""
import os
os.environ.setdefault("READ_HIP_DEBUG", "1")   # the probes live in libreadhip_debug.so only (python -m read_amd.build --debug)
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from read_amd import _lib  # noqa: E402

L = _lib.lib()
scratch = torch.zeros(256 * 8192, device="cuda")
cycles = torch.zeros(8192, dtype=torch.int64, device="cuda")
iters = 2000
NAMES = ["v_fma_f32", "v_fmac_f32 sgpr", "v_pk_fma_f32", "v_pk_fma_f32 sgpr pair + op_sel broadcast", "v_pk_add_f32", "v_pk_mul_f32",
         "v_pk_fma_f32 half-selects", "v_add_f32"]
for wps in (1, 2, 4):
    for mode in range(8):
        for rep in range(2):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            _lib.check(L.read_debug_valu_probe(mode, 256 * wps, iters, scratch.data_ptr(), cycles.data_ptr(), _lib.stream_ptr()))
            e1.record()
            e1.synchronize()
        ms = e0.elapsed_time(e1)
        per = float(cycles[:256 * wps * 4].double().mean()) / (iters * 64)
        print({"waves_per_simd": wps, "form": NAMES[mode], "ticks_per_instruction_per_simd": round(per / wps, 2),
               "ns_per_instruction_per_simd": round(1e6 * ms / (iters * 64 * wps), 3)}, flush=True)
