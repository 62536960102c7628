"""The 1x1 layers of the plan on the split-operand pixel-lane kernel (config -10) against the fp32 kernels' automatic choice
(conv_pxh = 0), one launch at a time, us per launch and effective GB/s (input + output + addend + residual bytes) — and, with the
debug library (READ_HIP_DEBUG=1), the kernel's attribution probes (read_tuning_set("conv_ablate", bits): 1 no epilogue memory
traffic, 2 activation loads from one resident line per lane, 4 no MFMAs, 16 no weight copy; results invalid).  Run on the GPU box."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from read_amd import _lib, synthetic                                   # noqa: E402
from read_amd.gated_conv import PackedGatedConv, gated_conv            # noqa: E402

H, W = 352, 1216
# (label, srcs [(C, shift)], cout, level, pre (C2, shift) or None, linear)
SHAPES = [
    ("Convs.2 64->32 @L0", [(32, 0), (32, 0)], 32, 0, None, False),
    ("Convs.1 128->64 @L1", [(64, 0), (64, 0)], 64, 1, None, False),
    ("Convs.0 256->128 @L2", [(128, 0), (128, 0)], 128, 2, None, False),
    ("AFFs.0.conv.0r 32->32 @L0 +pre", [(32, 0)], 32, 0, (64, 1), False),
    ("AFFs.1.conv.0r 96->64 @L1 +pre", [(32, 1), (64, 0)], 64, 1, (256, 1), False),
    ("AFFs.2.conv.0r 224->128 @L2 +pre", [(32, 2), (64, 1), (128, 0)], 128, 2, (448, 1), False),
    ("AFFq1 64->32 lin @L1 +pre", [(64, 0)], 32, 1, (192, 1), True),
    ("AFFq2 128->96 lin @L2 +pre", [(128, 0)], 96, 2, (448, 1), True),
    ("AFFq3 256->224 lin @L3", [(256, 0)], 224, 3, None, True),
    ("SCM2.main.1 16->32 @L1", [(16, 0)], 32, 1, None, False),
    ("SCM2.main.3 32->56 @L1", [(32, 0)], 56, 1, None, False),
    ("SCM2.conv 64->64 @L1 (8+56)", [(8, 0), (56, 0)], 64, 1, None, False),
    ("SCM1.main.3 64->120 @L2", [(64, 0)], 120, 2, None, False),
    ("SCM1.conv 128->128 @L2 (8+120)", [(8, 0), (120, 0)], 128, 2, None, False),
    ("SCM0.main.1 64->128 @L3", [(64, 0)], 128, 3, None, False),
    ("SCM0.main.3 128->248 @L3", [(128, 0)], 248, 3, None, False),
    ("SCM0.conv 256->256 @L3 (8+248)", [(8, 0), (248, 0)], 256, 3, None, False),
]
debug = os.environ.get("READ_HIP_DEBUG") == "1"
L = _lib.lib()
rows = []
for (label, srcs, cout, lvl, pre, linear) in SHAPES:
    h, w = H >> lvl, W >> lvl
    cin = sum(c for c, _ in srcs)
    st = synthetic.make_unet_state([("L", cin, cout, 1)], 1)
    b = "L.block."
    pk = PackedGatedConv(st[b + "conv_f.weight"], st[b + "conv_f.bias"], st[b + "conv_m.weight"], st[b + "conv_m.bias"], st[b + "norm.weight"],
                         st[b + "norm.bias"], st[b + "norm.running_mean"], st[b + "norm.running_var"], src_channels=[c for c, _ in srcs])
    xs, nbytes = [], 0
    for c, sh in srcs:
        hh, ww = (h << sh, w << sh) if sh > 0 else (h >> -sh, w >> -sh)
        xs.append((torch.randn(hh, ww, c, device="cuda"), sh))
        nbytes += min(hh * ww, h * w) * c * 4
    out = torch.empty(h, w, cout * (2 if linear else 1), device="cuda")
    nbytes += out.numel() * 4
    pr = None
    if pre is not None:
        pr = (torch.randn(h >> pre[1], w >> pre[1], pre[0], device="cuda"), 0, cout, pre[1])
        nbytes += (h >> pre[1]) * (w >> pre[1]) * 2 * cout * 4

    def timed(cfg, n=20):
        for _ in range(3):
            gated_conv(pk, xs, elu=True, config=cfg, out=out, pre=pr, linear=linear)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            gated_conv(pk, xs, elu=True, config=cfg, out=out, pre=pr, linear=linear)
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / n * 1e3

    row = {"layer": label, "MB": nbytes / 1e6, "pxh_us": timed(-10)}
    _lib.check(L.read_tuning_set(b"conv_pxh", 0))
    row["fp32_auto_us"] = timed(-1)
    _lib.check(L.read_tuning_set(b"conv_pxh", 16))
    if debug:
        for bits in (1, 2, 4, 16, 3, 7, 23):
            _lib.check(L.read_tuning_set(b"conv_ablate", bits))
            row[f"abl{bits}_us"] = timed(-10)
        _lib.check(L.read_tuning_set(b"conv_ablate", 0))
    rows.append(row)
    print("%-34s %6.1f MB  pxh %6.1f us (%5.0f GB/s)  fp32 %6.1f us  " % (label, row["MB"], row["pxh_us"], nbytes / row["pxh_us"] / 1e3, row["fp32_auto_us"]) +
          "  ".join("%s %.1f" % (k[:-3], v) for k, v in row.items() if k.startswith("abl")), flush=True)
if len(sys.argv) > 1:
    json.dump(rows, open(sys.argv[1], "w"), indent=1)
