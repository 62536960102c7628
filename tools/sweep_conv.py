"""Tile-configuration sweep of the gated-conv kernel on the UNet's real layer shapes (run on the GPU box).

    python tools/sweep_conv.py [--out gpurun_out/sweep_conv.json]
"""
import argparse
import json
import os
import re
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from read_amd import synthetic                                  # noqa: E402
from read_amd.gated_conv import PackedGatedConv, config_names, gated_conv   # noqa: E402

# (name, srcs [(C, shift)], cout, k, stride, outH, outW) at 1216x352
H, W = 352, 1216
SHAPES = [
    ("L0 32->32 3x3", [(32, 0)], 32, 3, 1, H, W),
    ("L1 64->64 3x3", [(64, 0)], 64, 3, 1, H // 2, W // 2),
    ("L2 128->128 3x3", [(128, 0)], 128, 3, 1, H // 4, W // 4),
    ("L3 256->256 3x3", [(256, 0)], 256, 3, 1, H // 8, W // 8),
    ("AFF0 480->32 1x1", [(32, 0), (64, -1), (128, -2), (256, -3)], 32, 1, 1, H, W),
    ("AFF1 480->64 1x1", [(32, 1), (64, 0), (128, -1), (256, -2)], 64, 1, 1, H // 2, W // 2),
    ("AFF2 480->128 1x1", [(32, 2), (64, 1), (128, 0), (256, -1)], 128, 1, 1, H // 4, W // 4),
    ("fe0 8->32 3x3", [(8, 0)], 32, 3, 1, H, W),
    ("fe1 32->64 3x3 s2", [(32, 0)], 64, 3, 2, H // 2, W // 2),
    ("fe2 64->128 3x3 s2", [(64, 0)], 128, 3, 2, H // 4, W // 4),
    ("fe6 128->256 3x3 s2", [(128, 0)], 256, 3, 2, H // 8, W // 8),
    ("fe4 64->32 4x4 s2", [(64, 0)], 32, 4, 2, H // 4, W // 4),
    ("fe3 128->64 4x4 s2", [(128, 0)], 64, 4, 2, H // 8, W // 8),
    ("fe7 256->128 4x4 s2", [(256, 0)], 128, 4, 2, H // 16, W // 16),
    ("Convs2 64->32 1x1", [(32, 0), (32, 0)], 32, 1, 1, H, W),
    ("fe5 32->3 3x3", [(32, 0)], 3, 3, 1, H, W),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/sweep_conv.json")
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--main-only", action="store_true", help="only the four dominant 3x3 shapes, automatic config")
    ap.add_argument("--only", default="", help="substring a config name must contain")
    ap.add_argument("--shapes", type=int, default=0, help="first N shapes only")
    ap.add_argument("--tune", default="", help="key=value[,key=value] for read_tuning_set")
    ap.add_argument("--fill", default="randn", help="input data: randn | zeros | ones (DVFS sensitivity)")
    a = ap.parse_args()
    names = config_names()
    if a.tune:
        from read_amd import _lib
        for kv in a.tune.split(","):
            k_, v_ = kv.split("=")
            _lib.check(_lib.lib().read_tuning_set(k_.encode(), int(v_)))
    res = []
    shapes = SHAPES[:4] if a.main_only else (SHAPES[:a.shapes] if a.shapes else SHAPES)
    for (label, srcs, cout, k, s, oh, ow) in shapes:
        cin = sum(c for c, _ in srcs)
        kc = 8 if any(c % 16 for c, _ in srcs) else 16
        st = synthetic.make_unet_state([("L", cin, cout, k)], 1)
        b = "L.block."
        pk = PackedGatedConv(st[b + "conv_f.weight"], st[b + "conv_f.bias"], st[b + "conv_m.weight"],
                             st[b + "conv_m.bias"], st[b + "norm.weight"], st[b + "norm.bias"],
                             st[b + "norm.running_mean"], st[b + "norm.running_var"], src_channels=[c for c, _ in srcs])
        pk32 = None
        if k == 1 and all(c % 32 == 0 for c, _ in srcs):           # 1x1 with 32-channel chunks: its own packing
            pk32 = PackedGatedConv(st[b + "conv_f.weight"], st[b + "conv_f.bias"], st[b + "conv_m.weight"],
                                   st[b + "conv_m.bias"], st[b + "norm.weight"], st[b + "norm.bias"],
                                   st[b + "norm.running_mean"], st[b + "norm.running_var"],
                                   src_channels=[c for c, _ in srcs], kc=32)
        ih, iw = oh * s, ow * s
        xs = []
        for c, sh in srcs:
            hh = (ih << sh) if sh > 0 else (ih >> -sh)
            ww = (iw << sh) if sh > 0 else (iw >> -sh)
            t = torch.randn(hh, ww, c, device="cuda")
            if a.fill == "zeros":
                t.zero_()
            elif a.fill == "ones":
                t.fill_(1.0)
            xs.append((t, sh))
        out = torch.empty(oh, ow, cout, device="cuda")
        groups = (cout + 31) // 32
        flops = 4.0 * oh * ow * cout * cin * k * k
        cand = [(-1, "auto")] if a.main_only else list(enumerate(names))
        for ci, name in cand:
            if ci < 0:
                for _ in range(a.iters + 2):
                    gated_conv(pk, xs, stride=s, elu=True, config=-1, out=out)
                torch.cuda.synchronize()
                continue
            m = re.match(r"k(\d)s(\d)c(\d+)_(?:wave_)?p(\d)q(\d)(?:m(\d)n(\d))?", name)
            ks, ss, kcc, P, QG, WM, WN = (int(g) if g is not None else 1 for g in m.groups())
            use32 = kcc == 32 and pk32 is not None and (ks, ss) == (k, s)
            if ((ks, ss, kcc) != (k, s, kc) and not use32) or groups % (WN * QG) or a.only not in name:
                continue
            pkc = pk32 if use32 else pk
            try:
                for _ in range(2):
                    gated_conv(pkc, xs, stride=s, elu=True, config=ci, out=out)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.iters):
                    gated_conv(pkc, xs, stride=s, elu=True, config=ci, out=out)
                e1.record()
                e1.synchronize()
                ms = e0.elapsed_time(e1) / a.iters
                r = {"shape": label, "config": ci, "name": name, "ms": ms, "tflops": flops / ms / 1e9}
            except Exception as ex:    # noqa: BLE001
                r = {"shape": label, "config": ci, "name": name, "error": str(ex)}
            print(r, flush=True)
            res.append(r)
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    with open(a.out, "w") as fh:
        json.dump(res, fh, indent=1)


if __name__ == "__main__":
    main()
