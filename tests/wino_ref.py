"""NumPy model of the Winograd F(2x2,3x3) gated-conv kernel (gated_conv_wino_kernel in read_amd/csrc/conv.hip): the filter
transform + fragment packing, the per-wave input transform, the MFMA contraction per frequency and the
cross-wave output transform — written with the SAME index maps as the HIP kernel so that layout mistakes
show up on the CPU.  Used by tests/test_wino_cpu.py against torch's conv2d."""
import numpy as np

BT = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], np.float32)
G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], np.float32)
AT = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], np.float32)


def filter_transform(w):
    """w (Cout,Cin,3,3) -> U (4,4,Cin,Cout): U[i][j] = (G g G^T)[i][j]."""
    return np.einsum("ia,ocab,jb->ijco", G, w.astype(np.float32), G).astype(np.float32)


def pack_wino(wf, wm):
    """-> flat float32 array ordered [group][k8 step][row i][j][f|m][lane 64][4], cout padded to 32;
    row 2 negated (the kernel forms d1 - d2 = -(B^T d)[2] so that every row is d[ra] +- d[rb])."""
    cout, cin = wf.shape[:2]
    cp = (cout + 31) // 32 * 32
    Uf, Um = filter_transform(wf), filter_transform(wm)
    out = np.zeros((cp // 32, cin // 8, 4, 4, 2, 64, 4), np.float32)
    lane = np.arange(64)
    for g in range(cp // 32):
        co = g * 32 + (lane & 31)
        ok = co < cout
        for s in range(cin // 8):
            for e in range(4):
                ci = 8 * s + 4 * (lane >> 5) + e
                for i in range(4):
                    for j in range(4):
                        sg = -1.0 if i == 2 else 1.0
                        out[g, s, i, j, 0, ok, e] = sg * Uf[i, j, ci[ok], co[ok]]
                        out[g, s, i, j, 1, ok, e] = sg * Um[i, j, ci[ok], co[ok]]
    return out.reshape(-1)


def wino_conv_model(x_hwc, packed, cin, cout):
    """Emulates the kernel: x (H,W,Cin) NHWC -> (f, m) pre-activation maps (H,W,Cout) each.
    Tile blocks of 4x8 tiles (8x16 output pixels); MFMA row t = tr*8 + tc; wave = frequency row i."""
    H, W, _ = x_hwc.shape
    cp = (cout + 31) // 32 * 32
    P = packed.reshape(cp // 32, cin // 8, 4, 4, 2, 64, 4)
    xp = np.zeros((H + 18, W + 18, cin), np.float32)
    xp[1:H + 1, 1:W + 1] = x_hwc                                  # origin shift: input row (oy-1) at index oy
    f = np.zeros((H, W, cp), np.float32)
    m = np.zeros((H, W, cp), np.float32)
    lane = np.arange(64)
    t = lane & 31
    tr, tc, half = t >> 3, t & 7, lane >> 5
    sign = {0: (0, 2, 1, -1), 1: (1, 2, 1, 1), 2: (1, 2, 1, -1), 3: (1, 3, 1, -1)}
    for by in range((H + 7) // 8):
        for bx in range((W + 15) // 16):
            oy0, ox0 = by * 8, bx * 16
            for g in range(cp // 32):
                R = np.zeros((4, 2, 2, 32, 32), np.float32)           # [row i][b][f|m][tile][cout]
                for i in range(4):
                    ra, rb, sa, sb = sign[i]
                    acc = np.zeros((4, 2, 32, 32), np.float32)        # [j][f|m][tile][cout]
                    for s in range(cin // 8):
                        # A fragments of the 64 lanes: V[j] is a float4 over cin 8s + 4*half + e
                        V = np.zeros((4, 64, 4), np.float32)
                        tt = np.zeros((4, 64, 4), np.float32)
                        for c in range(4):
                            ya, yb = oy0 + 2 * tr + ra, oy0 + 2 * tr + rb
                            xx = ox0 + 2 * tc + c
                            for e in range(4):
                                ci = 8 * s + 4 * half + e
                                tt[c, :, e] = sa * xp[ya, xx, ci] + sb * xp[yb, xx, ci]
                        V[0], V[1], V[2], V[3] = tt[0] - tt[2], tt[1] + tt[2], tt[2] - tt[1], tt[1] - tt[3]
                        for j in range(4):
                            for fm in range(2):
                                Bq = P[g, s, i, j, fm]                 # (64,4)
                                for e in range(4):
                                    # MFMA 32x32x2: A[row=lane&31][k=lane>>5], B[k=lane>>5][col=lane&31]
                                    A2 = np.zeros((32, 2), np.float32)
                                    B2 = np.zeros((2, 32), np.float32)
                                    A2[lane & 31, lane >> 5] = V[j][:, e]
                                    B2[lane >> 5, lane & 31] = Bq[:, e]
                                    acc[j, fm] += A2 @ B2
                    R[i, 0] = acc[0] + acc[1] + acc[2]
                    R[i, 1] = acc[1] - acc[2] - acc[3]
                for b in range(2):
                    for fm in range(2):
                        Y0 = R[0, b, fm] + R[1, b, fm] + R[2, b, fm]
                        Y1 = R[1, b, fm] - R[2, b, fm] - R[3, b, fm]
                        dst = m if fm else f
                        for tile in range(32):
                            ttr, ttc = tile >> 3, tile & 7
                            for a, Y in ((0, Y0), (1, Y1)):
                                oy, ox = oy0 + 2 * ttr + a, ox0 + 2 * ttc + b
                                if oy < H and ox < W:
                                    dst[oy, ox, g * 32:(g + 1) * 32] = Y[tile]
    return f[:, :, :cout], m[:, :, :cout]
