"""CPU numerics study for DESIGN.md 12.1 (c): the Winograd F(4x4,3x3) products on the bf16 matrix cores with SPLIT fp32 operands.

    python tools/bf16x3_study.py [--md profiles/r5_bf16x3_study.md]

The F(4x4) kernel multiplies transformed weights U (6 x 6 x Cin x Cout) with transformed patches V (6 x 6 x Cin x tiles) on
v_mfma_f32_16x16x4_f32, the slow matrix path of the chip (157 TF against 2.5 PF for bf16) and the one that shares the FP32 vector
pipe.  An fp32 number splits exactly into bf16 pieces: x = h + m + l (three pieces of 8 significant bits carry the 24 of an fp32);
a product of two bf16 numbers is exact in fp32, and the bf16 MFMA accumulates in fp32 — so U V = sum over piece pairs, and the
question is how many of the nine pairs are needed and what the result looks like against the kernel in the product.

This script answers it with numpy only (no GPU): one gated 3x3 layer (both convolutions, bias, ELU x sigmoid gate) at the UNet's
four widths, seeded activations and weights of the sizes the network sees, every variant against an fp64 direct convolution:

  fp32        the product kernel's arithmetic: V and U in fp32, products and sums in fp32 (numpy float32 matmul)
  bf16 x N    N piece pairs, largest first: hh | hm mh | mm hl lh | ml lm | ll; every pair one bf16 MFMA (exact products, fp32 sums)

It reports the relative error of the pre-activations and the PSNR of the gated output against fp64, next to the guard the test
suite holds the network to (>= 120 dB, tests/test_gpu_unet.py).  It says nothing about speed: splitting costs vector
instructions in the input transform (three pieces of 36 frequencies per (tile, channel)), counted in DESIGN.md 12.1 (c).
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from wino4_ref import AT, BT, G                                                     # noqa: E402


def bf16_round(x):
    """fp32 -> the nearest bf16 (ties to even), returned as fp32."""
    u = np.ascontiguousarray(x, np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32).reshape(np.shape(x))


def split3(x):
    h = bf16_round(x)
    m = bf16_round((x - h).astype(np.float32))
    l = bf16_round((x - h - m).astype(np.float32))
    return h, m, l


PAIRS = [(0, 0), (0, 1), (1, 0), (1, 1), (0, 2), (2, 0), (1, 2), (2, 1), (2, 2)]     # (piece of U, piece of V), by magnitude


def layer(C, hw, rng):
    H = W = hw
    x = rng.standard_normal((H + 2, W + 2, C)).astype(np.float32) * 0.6              # activations after ELU x sigmoid x BatchNorm: O(1)
    x[0], x[-1], x[:, 0], x[:, -1] = 0, 0, 0, 0                                      # zero padding
    k = 1.0 / np.sqrt(9 * C)
    w = [rng.uniform(-k, k, (C, C, 3, 3)).astype(np.float32) for _ in range(2)]      # conv_f, conv_m (nn.Conv2d default init)
    b = [rng.uniform(-k, k, C).astype(np.float32) for _ in range(2)]
    return x, w, b


def direct64(x, w):
    H, W = x.shape[0] - 2, x.shape[1] - 2
    out = np.zeros((H, W, w.shape[0]), np.float64)
    for ky in range(3):
        for kx in range(3):
            out += x[ky:ky + H, kx:kx + W].astype(np.float64) @ w[:, :, ky, kx].astype(np.float64).T
    return out


def tiles_of(x):
    """(H+2, W+2, C) -> raw 6 x 6 patches (ty, tx, 6, 6, C) of the 4 x 4 output tiles."""
    H, W = x.shape[0] - 2, x.shape[1] - 2
    ty, tx = H // 4, W // 4
    p = np.empty((ty, tx, 6, 6, x.shape[2]), np.float32)
    for i in range(6):
        for j in range(6):
            p[:, :, i, j] = x[i:i + 4 * ty:4, j:j + 4 * tx:4]
    return p


def wino(x, w, n_pairs):
    """n_pairs 0: fp32 products; else that many bf16 piece pairs.  -> (H, W, Cout) pre-activations without bias, fp32."""
    p = tiles_of(x)
    V = np.einsum("ia,yxabc,jb->ijyxc", BT, p, BT).astype(np.float32)               # the kernel: fp32 adds / fmas; same roundings to ~1 ulp
    U = np.einsum("ia,ocab,jb->ijco", G, w.astype(np.float64), G).astype(np.float32)   # host packer: fp64, rounded once
    ty, tx = p.shape[:2]
    M = np.zeros((6, 6, ty * tx, w.shape[0]), np.float32)
    if n_pairs == 0:
        for i in range(6):
            for j in range(6):
                M[i, j] = V[i, j].reshape(ty * tx, -1) @ U[i, j]
    else:
        Us, Vs = split3(U), split3(V)
        for i in range(6):
            for j in range(6):
                for (a, b_) in reversed(PAIRS[:n_pairs]):                            # small terms first, as a kernel would order its MFMAs
                    M[i, j] += Vs[b_][i, j].reshape(ty * tx, -1) @ Us[a][i, j]
    Y = np.einsum("ia,abtc,jb->tijc", AT, M, AT).astype(np.float32)                 # (tiles, 4, 4, Cout)
    return Y.reshape(ty, tx, 4, 4, -1).transpose(0, 2, 1, 3, 4).reshape(4 * ty, 4 * tx, -1)


def gate(f, m, bf, bm):
    f = f + bf
    m = m + bm
    return np.where(f > 0, f, np.expm1(np.minimum(f, 0))) / (1.0 + np.exp(-m))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--md", default="")
    a = ap.parse_args()
    rng = np.random.default_rng(5)
    rows = []
    for C, hw in ((32, 96), (64, 64), (128, 48), (256, 32)):
        x, w, b = layer(C, hw, rng)
        ref = [direct64(x, w[k]) for k in range(2)]
        gref = gate(ref[0], ref[1], b[0].astype(np.float64), b[1].astype(np.float64))
        scale = float(np.abs(np.concatenate([r.ravel() for r in ref])).max())
        for n in (0, 1, 3, 4, 6, 9):
            y = [wino(x, w[k], n) for k in range(2)]
            err = max(float(np.abs(y[k].astype(np.float64) - ref[k]).max()) for k in range(2)) / scale
            g = gate(y[0].astype(np.float32), y[1].astype(np.float32), b[0], b[1]).astype(np.float64)
            mse = float(np.mean((g - gref) ** 2))
            psnr = 10.0 * np.log10(float(np.abs(gref).max()) ** 2 / mse) if mse > 0 else float("inf")
            rows.append((C, hw, "fp32 products (the kernel)" if n == 0 else f"bf16 x {n} pairs", err, psnr))
            print("C=%3d %3dx%-3d %-28s max |err| / max |pre-activation| %.3e   gated output %.1f dB" % ((C, hw, hw) + rows[-1][2:]), flush=True)
    if a.md:
        with open(a.md, "w") as fh:
            fh.write("| C | image | products | max abs error / max pre-activation (vs fp64 direct) | gated output vs fp64, dB |\n|---|---|---|---|---|\n")
            for (C, hw, name, err, psnr) in rows:
                fh.write(f"| {C} | {hw}x{hw} | {name} | {err:.2e} | {psnr:.1f} |\n")


if __name__ == "__main__":
    main()
