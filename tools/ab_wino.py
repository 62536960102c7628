"""A/B of the Winograd kernels on the four dominant UNet shapes (run on the GPU box): "old" = F(2x2,3x3) row-per-wave kernel
(32x32x2 MFMA, cross-wave LDS epilogue), "new" = F(2x2,3x3) wave-autonomous kernel with the shared input transform (16x16x4 MFMA,
in-lane epilogue), "f4" = F(4x4,3x3).  --abl lists attribution variants of the f4 / new kernels (the -DREAD_DEBUG_KNOBS library:
python -m read_amd.build --debug, then READ_HIP_DEBUG=1).

    python tools/ab_wino.py [--iters 20] [--kernels old,new,f4] [--abl 1,7,24,...] [--tune key=value,...]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from read_amd import _lib, synthetic                                         # noqa: E402
from read_amd.gated_conv import PackedGatedConv, config_names, gated_conv    # noqa: E402

H, W = 352, 1216
SHAPES = [("L0 32", 32, H, W), ("L1 64", 64, H // 2, W // 2), ("L2 128", 128, H // 4, W // 4), ("L3 256", 256, H // 8, W // 8)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--tune", default="")
    ap.add_argument("--out", default="gpurun_out/ab_wino.json")
    ap.add_argument("--kernels", default="old,new,f4")
    ap.add_argument("--abl", default="", help="attribution probes of the f4 / new kernels (READ_HIP_DEBUG=1 library): comma list of bit sets")
    a = ap.parse_args()
    if a.tune:
        for kv in a.tune.split(","):
            k_, v_ = kv.split("=")
            _lib.check(_lib.lib().read_tuning_set(k_.encode(), int(v_)))
    old = [i for i, n in enumerate(config_names()) if "wino" in n][0]
    res = []
    for label, c, h, w in SHAPES:
        st = synthetic.make_unet_state([("L", c, c, 3)], 1)
        b = "L.block."
        pk = PackedGatedConv(st[b + "conv_f.weight"], st[b + "conv_f.bias"], st[b + "conv_m.weight"], st[b + "conv_m.bias"],
                             st[b + "norm.weight"], st[b + "norm.bias"], st[b + "norm.running_mean"], st[b + "norm.running_var"],
                             src_channels=[c])
        x = torch.randn(h, w, c, device="cuda")
        r = torch.randn(h, w, c, device="cuda")
        outs = {}
        for name, cfg in (("old", old), ("new", -3), ("f4", -5)):
            if name not in a.kernels.split(","):
                continue
            out = torch.empty(h, w, c, device="cuda")
            for _ in range(3):
                gated_conv(pk, [(x, 0)], elu=True, residual=r, config=cfg, out=out)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                gated_conv(pk, [(x, 0)], elu=True, residual=r, config=cfg, out=out)
            e1.record()
            e1.synchronize()
            ms = e0.elapsed_time(e1) / a.iters
            fl = 4.0 * h * w * c * c * 9
            outs[name] = out
            gain = 4.0 if name == "f4" else 2.25
            rec = {"shape": label, "kernel": name, "us": 1e3 * ms, "algorithmic_TF": fl / ms / 1e9, "executed_TF": fl / gain / ms / 1e9,
                   "frac_of_157.3": fl / gain / ms / 1e9 / 157.3}
            print(rec, flush=True)
            res.append(rec)
            for bits in [int(v) for v in a.abl.split(",") if v] if name in ("f4", "new") else []:
                _lib.check(_lib.lib().read_tuning_set(b"conv_abl", bits))
                for _ in range(2):
                    gated_conv(pk, [(x, 0)], elu=True, residual=r, config=cfg, out=out)
                e0.record()
                for _ in range(a.iters):
                    gated_conv(pk, [(x, 0)], elu=True, residual=r, config=cfg, out=out)
                e1.record()
                e1.synchronize()
                _lib.check(_lib.lib().read_tuning_set(b"conv_abl", 0))
                rec = {"shape": label, "kernel": name, "abl": bits, "us": 1e3 * e0.elapsed_time(e1) / a.iters}
                print(rec, flush=True)
                res.append(rec)
            gated_conv(pk, [(x, 0)], elu=True, residual=r, config=cfg, out=out)
        for other in ("new", "f4"):
            if "old" in outs and other in outs:
                d = (outs["old"] - outs[other]).abs().max().item()
                print({"shape": label, f"max_abs_diff_old_vs_{other}": d}, flush=True)
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
