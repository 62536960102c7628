"""Configuration sweep of the SMALL launches of the plan (SCM chains, levels 1-3): automatic choice against every compiled tile
configuration and the pixel-lane kernel (run on the GPU box).   python tools/sweep_small.py [--iters 20]"""
import argparse
import os
import re
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from read_amd import _lib, synthetic                                            # noqa: E402
from read_amd.gated_conv import PackedGatedConv, config_names, gated_conv        # noqa: E402

H, W = 352, 1216
SHAPES = []
for lvl, P in ((1, 64), (2, 128), (3, 256)):
    h, w = H >> lvl, W >> lvl
    SHAPES += [(f"SCM L{lvl} main.0 8->{P // 4} 3x3", [8], P // 4, 3, h, w),
               (f"SCM L{lvl} main.1 {P // 4}->{P // 2} 1x1", [P // 4], P // 2, 1, h, w),
               (f"SCM L{lvl} main.3 {P // 2}->{P - 8} 1x1", [P // 2], P - 8, 1, h, w),
               (f"SCM L{lvl} conv cat[8,{P - 8}]->{P} 1x1", [8, P - 8], P, 1, h, w)]


def timeit(fn, iters):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    names = config_names()
    for (label, srcs, cout, k, oh, ow) in SHAPES:
        cin = sum(srcs)
        st = synthetic.make_unet_state([("L", cin, cout, k)], 1)
        b = "L.block."
        args = [st[b + n] for n in ("conv_f.weight", "conv_f.bias", "conv_m.weight", "conv_m.bias", "norm.weight", "norm.bias",
                                     "norm.running_mean", "norm.running_var")]
        pk = PackedGatedConv(*args, src_channels=srcs)
        pk32 = PackedGatedConv(*args, src_channels=srcs, kc=32) if (k == 1 and all(c % 32 == 0 for c in srcs)) else None
        xs = [(torch.randn(oh, ow, c, device="cuda"), 0) for c in srcs]
        out = torch.empty(oh, ow, cout, device="cuda")
        kc = 8 if any(c % 16 for c in srcs) else 16
        groups = (cout + 31) // 32
        res = {"auto": timeit(lambda: gated_conv(pk32 or pk, xs, elu=True, config=-1, out=out), a.iters)}
        for ci, name in enumerate(names):
            m = re.match(r"k(\d)s(\d)c(\d+)_(?:wave_)?p(\d)q(\d)(?:m(\d)n(\d))?", name)
            ks, ss, kcc, P, QG, WM, WN = (int(g) if g is not None else 1 for g in m.groups())
            use32 = kcc == 32 and pk32 is not None
            if (ks, ss) != (k, 1) or (kcc != kc and not use32) or groups % (WN * QG):
                continue
            try:
                res[name] = timeit(lambda: gated_conv(pk32 if use32 else pk, xs, elu=True, config=ci, out=out), a.iters)
            except _lib.ReadHipError:
                pass
        if k == 1:
            try:
                res["px(-2)"] = timeit(lambda: gated_conv(pk, xs, elu=True, config=-2, out=out), a.iters)
            except _lib.ReadHipError:
                pass
        best = min(res, key=res.get)
        print(f"{label:36s} auto {res['auto']:6.1f} us   best {best:28s} {res[best]:6.1f} us   " +
              " ".join(f"{n.split('_', 1)[-1] if '_' in n else n}={v:.1f}" for n, v in sorted(res.items(), key=lambda t: t[1])[:5]), flush=True)


if __name__ == "__main__":
    main()
