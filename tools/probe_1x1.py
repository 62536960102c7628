"""The 1x1 layers of the plan, one by one: every kernel that can run each (the table's LDS-tiled / wave-autonomous configs, the
pixel-lane kernel = config -2, the automatic choice = -1), us per launch and effective HBM GB/s (input + output + addend bytes).
Run on the GPU box."""
import os
import re
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from read_amd import synthetic                                # noqa: E402
from read_amd.gated_conv import PackedGatedConv, config_names, gated_conv   # noqa: E402

H, W = 352, 1216
# (label, srcs [(C, shift)], cout, level, pre (C2, shift) or None, residual)
SHAPES = [
    ("AFFs.0.conv.0r 32->32 @L0 +pre", [(32, 0)], 32, 0, (64, 1)),
    ("AFFs.1.conv.0r 96->64 @L1 +pre", [(32, 1), (64, 0)], 64, 1, (256, 1)),
    ("AFFs.2.conv.0r 224->128 @L2 +pre", [(32, 2), (64, 1), (128, 0)], 128, 2, (448, 1)),
    ("Convs.2 64->32 @L0", [(32, 0), (32, 0)], 32, 0, None),
    ("Convs.1 128->64 @L1", [(64, 0), (64, 0)], 64, 1, None),
    ("Convs.0 256->128 @L2", [(128, 0), (128, 0)], 128, 2, None),
    ("SCM2.conv 64->64 @L1 (8+56)", [(8, 0), (56, 0)], 64, 1, None),
    ("SCM2.main.3 32->56 @L1", [(32, 0)], 56, 1, None),
    ("SCM2.main.1 16->32 @L1", [(16, 0)], 32, 1, None),
]
names = config_names()
for (label, srcs, cout, lvl, pre) in SHAPES:
    h, w = H >> lvl, W >> lvl
    cin = sum(c for c, _ in srcs)
    st = synthetic.make_unet_state([("L", cin, cout, 1)], 1)
    b = "L.block."
    args = (st[b + "conv_f.weight"], st[b + "conv_f.bias"], st[b + "conv_m.weight"], st[b + "conv_m.bias"], st[b + "norm.weight"],
            st[b + "norm.bias"], st[b + "norm.running_mean"], st[b + "norm.running_var"])
    pk = PackedGatedConv(*args, src_channels=[c for c, _ in srcs])
    pk32 = PackedGatedConv(*args, src_channels=[c for c, _ in srcs], kc=32) if all(c % 32 == 0 for c, _ in srcs) else None
    xs = []
    nbytes = 0
    for c, sh in srcs:
        hh, ww = (h << sh, w << sh) if sh > 0 else (h >> -sh, w >> -sh)
        xs.append((torch.randn(hh, ww, c, device="cuda"), sh))
        nbytes += h * w * c * 4                                # what the layer needs of it
    out = torch.empty(h, w, cout, device="cuda")
    nbytes += h * w * cout * 4
    pr = None
    if pre is not None:
        pr = (torch.randn(h >> pre[1], w >> pre[1], pre[0], device="cuda"), 0, cout, pre[1])
        nbytes += (h >> pre[1]) * (w >> pre[1]) * 2 * cout * 4
    kc = 8 if any(c % 16 for c, _ in srcs) else 16
    groups = (cout + 31) // 32
    cands = [(-1, "auto", pk), (-2, "pixel-lane", pk)]
    for ci, name in enumerate(names):
        m = re.match(r"k(\d)s(\d)c(\d+)_(?:wave_)?p(\d)q(\d)(?:m(\d)n(\d))?", name)
        if not m:
            continue
        ks, ss, kcc, P, QG, WM, WN = (int(g) if g is not None else 1 for g in m.groups())
        if (ks, ss) != (1, 1) or groups % (WN * QG):
            continue
        if kcc == kc:
            cands.append((ci, name, pk))
        elif kcc == 32 and pk32 is not None:
            cands.append((ci, name, pk32))
    res = []
    for ci, name, p_ in cands:
        try:
            for _ in range(3):
                gated_conv(p_, xs, elu=True, config=ci, out=out, pre=pr)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                gated_conv(p_, xs, elu=True, config=ci, out=out, pre=pr)
            e1.record()
            e1.synchronize()
            us = e0.elapsed_time(e1) / 20 * 1e3
            res.append((us, name))
        except Exception as ex:    # noqa: BLE001
            res.append((float("inf"), name + " -> " + str(ex)[:60]))
    res.sort()
    print("%-36s %6.1f MB  " % (label, nbytes / 1e6) + "  ".join("%s %.1f us (%.0f GB/s)" % (n, u, nbytes / u / 1e3) for u, n in res if u < 1e9), flush=True)
    for u, n in res:
        if u == float("inf"):
            print("      ", n)
