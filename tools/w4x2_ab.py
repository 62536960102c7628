"""A/B of the F(4x4) kernels on the four dominant C->C shapes at 1216x352 (run on the GPU box).

    python tools/w4x2_ab.py [--knobs conv_w4x2=0 conv_w4x2=1 ...] [--iters 20]
Each knob set is a comma list for read_tuning_set; prints us per launch per level."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from read_amd import _lib, synthetic                            # noqa: E402
from read_amd.gated_conv import PackedGatedConv, gated_conv     # noqa: E402

H, W = 352, 1216


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--knobs", nargs="*", default=["conv_w4x2=0", "conv_w4x2=1"])
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--residual", type=int, default=1)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    L = _lib.lib()
    res = {}
    for lvl, c in enumerate((32, 64, 128, 256)):
        h, w = H >> lvl, W >> lvl
        st = synthetic.make_unet_state([("L", c, c, 3)], 1)
        b = "L.block."
        pk = PackedGatedConv(st[b + "conv_f.weight"], st[b + "conv_f.bias"], st[b + "conv_m.weight"], st[b + "conv_m.bias"],
                             st[b + "norm.weight"], st[b + "norm.bias"], st[b + "norm.running_mean"], st[b + "norm.running_var"],
                             src_channels=[c])
        x = torch.randn(h, w, c, device="cuda")
        r = torch.randn(h, w, c, device="cuda") if a.residual else None
        out = torch.empty(h, w, c, device="cuda")
        base = None
        for ks in a.knobs:
            sets = [kv.split("=") for kv in ks.split(",") if kv]
            for k_, v_ in sets:
                _lib.check(L.read_tuning_set(k_.encode(), int(v_)))
            for _ in range(3):
                gated_conv(pk, [(x, 0)], elu=True, residual=r, config=-5, out=out)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                gated_conv(pk, [(x, 0)], elu=True, residual=r, config=-5, out=out)
            e1.record()
            e1.synchronize()
            us = e0.elapsed_time(e1) / a.iters * 1e3
            o = out.clone()
            if base is None:
                base = o
            d = float((o - base).abs().max())
            res[f"C{c} {ks}"] = us
            print(f"C={c:3d} {ks:32s} {us:8.2f} us   maxdiff vs first {d:.3g}", flush=True)
            for k_, v_ in sets:
                _lib.check(L.read_tuning_set(k_.encode(), 0)) if k_ == "conv_w4x2" else None
    if a.out:
        with open(a.out, "w") as fh:
            json.dump(res, fh, indent=1)


if __name__ == "__main__":
    main()
