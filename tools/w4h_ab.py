"""A/B of the F(4x4) kernels on the four dominant C->C shapes at 1216x352 (run on the GPU box): fp32 matrix cores (config -5)
against split operands on the f16 matrix cores (config -7), both against the torch-fp32 oracle on the CPU.

    python tools/w4h_ab.py [--iters 20] [--check 1] [--out gpurun_out/r6_w4h_ab.json]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import unet_torch                                   # noqa: E402  (checker only)
from read_amd import synthetic                                  # noqa: E402
from read_amd.gated_conv import PackedGatedConv, gated_conv     # noqa: E402

H, W = 352, 1216


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--check", type=int, default=1)
    ap.add_argument("--out", default="")
    ap.add_argument("--levels", default="0,1,2,3")
    ap.add_argument("--s2", type=int, default=0, help="1: the three 3x3 / stride-2 layers instead")
    ap.add_argument("--mul", type=int, default=0, help="1: the FAM form x1 + BC(x1 * x2) (the kernels' MUL variants)")
    ap.add_argument("--cfg", type=int, default=-7, help="kernel the probes run on: -7 the Winograd split-operand kernel, -8 the direct one")
    ap.add_argument("--waves", type=int, default=4, help="kernel the probes run on: 4 (the product kernel) / 8 specialised waves")
    ap.add_argument("--abl", default="", help="comma list of conv_abl probe values for the split-operand kernel (READ_HIP_DEBUG=1; results invalid)")
    a = ap.parse_args()
    res = {}
    torch.manual_seed(0)
    if a.s2:                                                   # the three stride-2 layers: fp32 direct kernels against the split-operand one
        from read_amd import _lib as _l2
        for lvl in (0, 1, 2):
            cin, cout, h, w = 32 << lvl, 64 << lvl, H >> lvl, W >> lvl
            st = synthetic.make_unet_state([("L", cin, cout, 3)], 1)
            b = "L.block."
            pk = PackedGatedConv(st[b + "conv_f.weight"], st[b + "conv_f.bias"], st[b + "conv_m.weight"], st[b + "conv_m.bias"],
                                 st[b + "norm.weight"], st[b + "norm.bias"], st[b + "norm.running_mean"], st[b + "norm.running_var"],
                                 src_channels=[cin])
            xc = torch.randn(cin, h, w)
            x = xc.permute(1, 2, 0).contiguous().cuda()
            out = torch.empty(h // 2, w // 2, cout, device="cuda")
            with torch.no_grad():
                ref = unet_torch.basic_conv(st, "L", xc[None], 3, stride=2, elu=True)[0].permute(1, 2, 0)
            for name, knob in (("fp32 direct", 0), ("split direct", 32)):
                _l2.check(_l2.lib().read_tuning_set(b"conv_d3h_s2", knob))
                for _ in range(3):
                    gated_conv(pk, [(x, 0)], stride=2, elu=True, out=out)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.iters):
                    gated_conv(pk, [(x, 0)], stride=2, elu=True, out=out)
                e1.record()
                e1.synchronize()
                d = out.cpu() - ref
                print(f"s2 {cin:3d}->{cout:3d} {h}x{w} {name:12s} {e0.elapsed_time(e1) / a.iters * 1e3:8.2f} us   max |diff| {float(d.abs().max()):.3e}   "
                      f"{10.0 * torch.log10(ref.abs().max().double() ** 2 / float((d.double() ** 2).mean())).item():.1f} dB", flush=True)
            _l2.check(_l2.lib().read_tuning_set(b"conv_d3h_s2", 32))
        return
    for lvl in [int(v) for v in a.levels.split(",")]:
        c = 32 << lvl
        h, w = H >> lvl, W >> lvl
        st = synthetic.make_unet_state([("L", c, c, 3)], 1)
        b = "L.block."
        pk = PackedGatedConv(st[b + "conv_f.weight"], st[b + "conv_f.bias"], st[b + "conv_m.weight"], st[b + "conv_m.bias"],
                             st[b + "norm.weight"], st[b + "norm.bias"], st[b + "norm.running_mean"], st[b + "norm.running_var"],
                             src_channels=[c])
        xc, rc = torch.randn(c, h, w), torch.randn(c, h, w)
        x, r = xc.permute(1, 2, 0).contiguous().cuda(), rc.permute(1, 2, 0).contiguous().cuda()
        out = torch.empty(h, w, c, device="cuda")
        x2c = torch.randn(c, h, w) if a.mul else None
        x2 = x2c.permute(1, 2, 0).contiguous().cuda() if a.mul else None
        ref = None
        if a.check:
            with torch.no_grad():
                ref = (unet_torch.basic_conv(st, "L", (xc * x2c if a.mul else xc)[None], 3, elu=True)[0] + rc).permute(1, 2, 0)
        from read_amd import _lib as _l
        dbg = bool(os.environ.get("READ_HIP_DEBUG"))                 # the specialised-wave kernel lives in the debug library only
        for name, cfg in (("fp32", -5), ("f16x3", -7), ("d3h", -8)) + ((("f16x3w8", -7),) if dbg else ()):
            if cfg == -7 and a.mul:
                continue                                     # the Winograd split-operand kernel does not take FAM's multiply
            if dbg:
                _l.check(_l.lib().read_tuning_set(b"conv_w4h_waves", 8 if name.endswith("w8") else 4))
            for _ in range(3):
                gated_conv(pk, [(x, 0)], elu=True, residual=r, config=cfg, out=out, mul=x2)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                gated_conv(pk, [(x, 0)], elu=True, residual=r, config=cfg, out=out, mul=x2)
            e1.record()
            e1.synchronize()
            us = e0.elapsed_time(e1) / a.iters * 1e3
            line = f"C={c:3d} {h}x{w} {name:6s} {us:8.2f} us"
            rec = {"us": us}
            if ref is not None:
                d = (out.cpu() - ref)
                mse = float((d.double() ** 2).mean())
                rec["max_abs"] = float(d.abs().max())
                rec["psnr_db"] = 10.0 * torch.log10(ref.abs().max().double() ** 2 / mse).item()
                line += f"   max |diff| vs torch fp32 {rec['max_abs']:.3e}   {rec['psnr_db']:.1f} dB"
            res[f"C{c} {name}"] = rec
            print(line, flush=True)
        if a.abl:
            from read_amd import _lib
            if a.cfg == -7:
                _lib.check(_lib.lib().read_tuning_set(b"conv_w4h_waves", a.waves))
            for v in [int(t) for t in a.abl.split(",")]:
                _lib.check(_lib.lib().read_tuning_set(b"conv_abl", v))
                for _ in range(3):
                    gated_conv(pk, [(x, 0)], elu=True, residual=r, config=a.cfg, out=out, mul=x2)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.iters):
                    gated_conv(pk, [(x, 0)], elu=True, residual=r, config=a.cfg, out=out, mul=x2)
                e1.record()
                e1.synchronize()
                us = e0.elapsed_time(e1) / a.iters * 1e3
                res[f"C{c} f16x3 abl {v}"] = {"us": us}
                print(f"C={c:3d} f16x3 probe {v:5d} {us:8.2f} us", flush=True)
            _lib.check(_lib.lib().read_tuning_set(b"conv_abl", 0))
    if a.out:
        with open(a.out, "w") as fh:
            json.dump(res, fh, indent=1)


if __name__ == "__main__":
    main()
