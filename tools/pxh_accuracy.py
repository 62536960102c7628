"""Accuracy of the split-operand 1x1 / implicit-GEMM kernel against a float64 convolution, next to the fp32 kernels on the same layer
(run on the GPU box): relative rms error of the pre-activations (linear launches: conv_f + b_f) and PSNR of the gated output."""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from read_amd import _lib, synthetic                                   # noqa: E402
from read_amd.gated_conv import PackedGatedConv, gated_conv            # noqa: E402

L = _lib.lib()
torch.manual_seed(0)
print("| layer | kernel | rel. rms of conv_f against float64 | gated output PSNR against float64 (dB) |")
print("|---|---|---|---|")
for (cin, cout, k, H, W) in ((64, 32, 1, 96, 256), (256, 128, 1, 88, 304), (480 // 2, 64, 1, 44, 152), (8, 32, 3, 176, 608)):
    st = synthetic.make_unet_state([("L", cin, cout, k)], 3)
    b = "L.block."
    pk = PackedGatedConv(st[b + "conv_f.weight"], st[b + "conv_f.bias"], st[b + "conv_m.weight"], st[b + "conv_m.bias"], st[b + "norm.weight"],
                         st[b + "norm.bias"], st[b + "norm.running_mean"], st[b + "norm.running_var"], src_channels=[cin])
    x = torch.randn(cin, H, W)
    xd = x[None].double()
    t = lambda n: torch.as_tensor(np.asarray(st[b + n])).double()       # noqa: E731
    f64 = F.conv2d(xd, t("conv_f.weight"), t("conv_f.bias"), padding=k // 2)[0]
    m64 = F.conv2d(xd, t("conv_m.weight"), t("conv_m.bias"), padding=k // 2)[0]
    y64 = F.batch_norm((F.elu(f64) * torch.sigmoid(m64))[None], t("norm.running_mean"), t("norm.running_var"), t("norm.weight"), t("norm.bias"),
                       training=False, eps=1e-5)[0]
    xs = [(x.permute(1, 2, 0).contiguous().cuda(), 0)]
    for name, knobs in (("split operands, f16 matrix cores", {}), ("fp32 matrix cores", {b"conv_pxh": 0, b"conv_t3h": 0})):
        for kk, v in knobs.items():
            _lib.check(L.read_tuning_set(kk, v))
        lin = gated_conv(pk, xs, linear=True).cpu().permute(2, 0, 1)[:cout].double()
        y = gated_conv(pk, xs, elu=True).cpu().permute(2, 0, 1).double()
        rel = float(((lin - f64) ** 2).mean().sqrt() / (f64 ** 2).mean().sqrt())
        mse = float(((y - y64) ** 2).mean())
        psnr = 10.0 * np.log10(float(y64.abs().max()) ** 2 / mse)
        print(f"| {cin} -> {cout}, {k}x{k}, {H}x{W} | {name} | {rel:.2e} | {psnr:.1f} |", flush=True)
        _lib.check(L.read_tuning_set(b"conv_pxh", 16))
        _lib.check(L.read_tuning_set(b"conv_t3h", 8))
