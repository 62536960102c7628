"""Ablation of the workgroup-tiled gated-conv kernel (read_tuning_set("conv_ablate", bits)); GPU box."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from read_amd import _lib, synthetic                                        # noqa: E402
from read_amd.gated_conv import PackedGatedConv, config_names, gated_conv   # noqa: E402

SHAPES = {"L0": (32, 352, 1216), "L1": (64, 176, 608), "L2": (128, 88, 304), "L3": (256, 44, 152)}
CASES = [("L0", "k3s1c16_p2q1m4n1f1b2"), ("L0", "k3s1c16_p1q1m4n1f2b1"), ("L2", "k3s1c16_p2q2m2n2f1b2"), ("L3", "k3s1c16_p2q1m2n2f2b1")]
BITS = [(0, "full"), (1, "no epilogue mem"), (2, "no A restage"), (4, "B pinned"), (8, "no MFMA"), (7, "only MFMA+LDS reads"),
        (15, "skeleton"), (14, "epilogue only"), (9, "loads only (A+B), no epi mem")]

L = _lib.lib()
names = config_names()
res = []
for shape, cname in CASES:
    C, H, W = SHAPES[shape]
    st = synthetic.make_unet_state([("L", C, C, 3)], 1)
    b = "L.block."
    pk = PackedGatedConv(st[b + "conv_f.weight"], st[b + "conv_f.bias"], st[b + "conv_m.weight"], st[b + "conv_m.bias"],
                         st[b + "norm.weight"], st[b + "norm.bias"], st[b + "norm.running_mean"], st[b + "norm.running_var"])
    x, r = torch.randn(H, W, C, device="cuda"), torch.randn(H, W, C, device="cuda")
    out = torch.empty(H, W, C, device="cuda")
    ci = names.index(cname)
    flops = 4.0 * H * W * C * C * 9
    for bits, label in BITS:
        _lib.check(L.read_tuning_set(b"conv_ablate", bits))
        for _ in range(3):
            gated_conv(pk, [(x, 0)], elu=True, residual=r, config=ci, out=out)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            gated_conv(pk, [(x, 0)], elu=True, residual=r, config=ci, out=out)
        e1.record()
        e1.synchronize()
        ms = e0.elapsed_time(e1) / 10
        row = {"shape": shape, "config": cname, "bits": bits, "what": label, "us": ms * 1e3, "tflops_equiv": flops / ms / 1e9}
        print("%-3s %-24s %2d %-32s %7.1f us  (%5.1f TF-equiv)" % (shape, cname, bits, label, ms * 1e3, flops / ms / 1e9), flush=True)
        res.append(row)
_lib.check(L.read_tuning_set(b"conv_ablate", 0))
json.dump(res, open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/ablate.json", "w"), indent=1)
