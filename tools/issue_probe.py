"""Issue model of gfx950 around fp32 MFMAs (run on the GPU box): cycles per MFMA slot as a function of the MFMA shape, the
number / kind of filler instructions in its shadow, and the number of waves per SIMD.

    python tools/issue_probe.py [out.json]
"""
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
gsrc = torch.ones(4096, device="cuda")
cycles = torch.zeros(4096, dtype=torch.int64, device="cuda")
FT = {0: "v_add indep", 1: "ds_read_b128", 2: "s_add", 3: "v_add chain", 4: "global_load_dwordx4", 5: "v_pk_add_f32", 6: "v_exp_f32",
      7: "ds_write_b128", 8: "v_fma_f32"}
ONLY = [int(v) for v in os.environ.get("PROBE_FILLERS", "0,1,2,3,4,5,6,7,8").split(",")]
res = []
iters = 2000
KINDS = [int(v) for v in os.environ.get("PROBE_KINDS", "0,1").split(",")]      # 2: v_mfma_f32_16x16x32_bf16 (round 5)
for kind in KINDS:
    for wps in (1, 2):                      # waves per SIMD = workgroups per CU
        blocks = 256 * wps
        for ft in ONLY:
            for K in (0, 1, 2, 3, 4, 6, 8, 12):
                if K == 0 and ft != ONLY[0]:
                    continue
                for rep in range(2):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    _lib.check(L.read_debug_issue_probe(kind, ft, K, blocks, iters, scratch.data_ptr(), cycles.data_ptr(), gsrc.data_ptr(),
                                                        _lib.stream_ptr()))
                    e1.record()
                    e1.synchronize()
                ms = e0.elapsed_time(e1)
                cyc = cycles[:blocks * 4].double()
                per = float(cyc.mean()) / (iters * 16)          # s_memtime ticks per MFMA slot of ONE wave
                row = {"mfma": ("32x32x2", "16x16x4", "16x16x32_bf16")[kind], "waves_per_simd": wps, "filler": FT[ft], "K": K, "ms": ms,
                       "ticks_per_mfma_per_wave": per, "ticks_per_mfma_per_simd": per / wps,
                       "ns_per_mfma_per_simd": 1e6 * ms / (iters * 16 * wps)}
                print(row, flush=True)
                res.append(row)
json.dump(res, open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/issue_probe.json", "w"), indent=1)
