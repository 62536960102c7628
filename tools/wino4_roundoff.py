"""fp32 round-off of Winograd F(4x4,3x3) against F(2x2,3x3) and the direct form on the UNet of the render path (CPU, torch).

DESIGN.md §3.3 / §12: before F(4x4,3x3) is built for the layers with C >= 128, how much of the 0.5 dB budget (north_star) and of
the tests' tolerance (max|diff| <= 5e-6, PSNR >= 120 dB) would its transforms cost?  The 3x3 / stride-1 convolutions of the
oracle network (oracle/unet_torch.py) are replaced by fp32 Winograd emulations — input, filter and output transforms and the
per-frequency channel contraction all in fp32, as the MFMA kernels compute them — and the RGB output is compared with the
same network evaluated in float64.

    python tools/wino4_roundoff.py [--width 1216 --height 352]
"""
import argparse
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import unet_torch                      # noqa: E402  (a measurement tool, like the tests: not product code)
from read_amd import synthetic                     # noqa: E402
from tests.unet_spec import UNET_SPEC              # noqa: E402

MATS = {
    2: (np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], np.float64),
        np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], np.float64),
        np.array([[1, 1, 1, 0], [0, 1, -1, -1]], np.float64)),
    4: (np.array([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0],
                  [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], np.float64),
        np.array([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6],
                  [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], np.float64),
        np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], np.float64)),
}


def wino_conv3x3(x, w, bias, m):
    """x (1,C,H,W), w (O,C,3,3) -> (1,O,H,W), zero padding 1, F(m x m, 3x3) with every step in x.dtype."""
    BT, G, AT = (torch.tensor(a, dtype=x.dtype) for a in MATS[m])
    _, C, H, W = x.shape
    ty, tx = -(-H // m), -(-W // m)
    xp = F.pad(x, (1, tx * m + 1 - W, 1, ty * m + 1 - H))
    d = xp.unfold(2, m + 2, m).unfold(3, m + 2, m)[0]                       # (C, ty, tx, m+2, m+2)
    U = torch.einsum("ia,ocab,jb->ijoc", G, w, G)                            # filter transform (on the host in the product)
    V = torch.einsum("ia,cyxab,jb->ijcyx", BT, d, BT)
    M = torch.einsum("ijoc,ijcyx->ijoyx", U, V)
    Y = torch.einsum("pi,ijoyx,qj->oypxq", AT, M, AT).reshape(w.shape[0], ty * m, tx * m)
    return (Y[:, :H, :W] + bias[:, None, None])[None]


def run(state, xs, dtype, policy):
    """policy(cin) -> 0 direct, 2 or 4: Winograd tile size for a 3x3 / stride-1 layer with cin input channels."""
    conv2d = F.conv2d

    def patched(x, w, b=None, stride=1, padding=0, *a, **k):
        if w.shape[-1] == 3 and stride == 1 and w.shape[1] % 16 == 0 and policy(w.shape[1]):
            return wino_conv3x3(x, w, b, policy(w.shape[1]))
        return conv2d(x, w, b, stride, padding, *a, **k)
    old_t = unet_torch._t
    unet_torch._t = lambda v: torch.as_tensor(np.asarray(v)).to(dtype)
    F.conv2d = patched
    try:
        with torch.no_grad():
            return unet_torch.unet_forward(state, *[x.to(dtype) for x in xs]).double()
    finally:
        F.conv2d = conv2d
        unet_torch._t = old_t


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=1216)
    ap.add_argument("--height", type=int, default=352)
    a = ap.parse_args()
    torch.manual_seed(0)
    state = synthetic.make_unet_state(UNET_SPEC)
    xs = [torch.rand(1, 8, a.height >> l, a.width >> l) for l in range(4)]
    # self-check of the emulation
    x, w, b = torch.randn(1, 16, 13, 21, dtype=torch.float64), torch.randn(8, 16, 3, 3, dtype=torch.float64), torch.randn(8, dtype=torch.float64)
    for m in (2, 4):
        assert float((wino_conv3x3(x, w, b, m) - F.conv2d(x, w, b, padding=1)).abs().max()) < 1e-10
    ref = run(state, xs, torch.float64, lambda c: 0)
    peak = float(ref.abs().max())
    print(f"{a.width}x{a.height}, output range +-{peak:.3f}, std {float(ref.std()):.4f}")
    for name, pol in (("direct fp32", lambda c: 0), ("F(2x2) everywhere (the product)", lambda c: 2),
                      ("F(4x4) for C >= 128, F(2x2) below", lambda c: 4 if c >= 128 else 2), ("F(4x4) everywhere", lambda c: 4)):
        out = run(state, xs, torch.float32, pol)
        err = out - ref
        mse = float((err ** 2).mean())
        print(f"  {name:36s} max|diff| {float(err.abs().max()):.3e}   PSNR(peak 1) {10 * np.log10(1.0 / mse):6.1f} dB   "
              f"rms/std {float(err.pow(2).mean().sqrt() / ref.std()):.2e}")


if __name__ == "__main__":
    main()
