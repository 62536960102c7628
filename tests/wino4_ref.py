"""NumPy model of the Winograd F(4x4,3x3) kernel on v_mfma_f32_16x16x4_f32 (gated_conv_wino4_kernel in read_amd/csrc/conv.hip),
with the kernel's index maps: filter transform + fragment order, thread -> (tile, input channel) of the shared input
transform and the swizzled V buffer it fills, MFMA operand / result lane maps, in-lane output transform, pixel / channel of
every result.  tests/test_wino_cpu.py compares it with torch's conv2d.

Unit = 2 x 8 tiles of 4 x 4 output pixels (8 x 32 pixels) x 32 output channels; wave w owns channels 8w .. 8w+7 (conv_f in MFMA
rows 0..7, conv_m in rows 8..15), the 16 tiles are the MFMA columns; 36 frequencies x 4 registers = 144 accumulators.
"""
import numpy as np

BT = np.array([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0],
               [0, 4, 0, -5, 0, 1]], np.float32)
G = np.array([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6],
              [0, 0, 1]], np.float64)
AT = np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], np.float32)
LANE = np.arange(64)


def filter_transform4(w):
    """w (Cout,Cin,3,3) -> U (6,6,Cin,Cout), computed in float64 and rounded once (as the host packer does)."""
    return np.einsum("ia,ocab,jb->ijco", G, w.astype(np.float64), G).astype(np.float32)


def pack_w4(wf, wm):
    """-> flat fp32 [group][wave 4][chunk of 16 cin][frequency 36 = 6 xi + nu][lane 64][e 4]; lane (i = lane & 15, kl = lane >> 4)
    holds U_{f if i < 8 else m}[xi][nu][cin = 16 chunk + 4 kl + e][cout = 32 g + 8 w + (i & 7)]."""
    cout, cin = wf.shape[:2]
    cp = (cout + 31) // 32 * 32
    U = [filter_transform4(wf), filter_transform4(wm)]
    out = np.zeros((cp // 32, 4, cin // 16, 36, 64, 4), np.float32)
    i, kl = LANE & 15, LANE >> 4
    for g in range(cp // 32):
        for w in range(4):
            co = 32 * g + 8 * w + (i & 7)
            ok = co < cout
            for c in range(cin // 16):
                for e in range(4):
                    ci = 16 * c + 4 * kl + e
                    for fm in range(2):
                        sel = ok & ((i >> 3) == fm)
                        for fq in range(36):
                            out[g, w, c, fq, sel, e] = U[fm][fq // 6, fq % 6, ci[sel], co[sel]]
    return out.reshape(-1)


def mfma_16x16x4(a_lane, b_lane):
    A = np.zeros((16, 4), np.float32)
    B = np.zeros((4, 16), np.float32)
    A[LANE & 15, LANE >> 4] = a_lane
    B[LANE >> 4, LANE & 15] = b_lane
    D = A @ B
    out = np.zeros((64, 4), np.float32)
    for r in range(4):
        out[:, r] = D[4 * (LANE >> 4) + r, LANE & 15]
    return out


def wino4_conv_model(x_hwc, packed, cin, cout):
    H, W, _ = x_hwc.shape
    cp = (cout + 31) // 32 * 32
    P = packed.reshape(cp // 32, 4, cin // 16, 36, 64, 4)
    xp = np.zeros((H + 20, W + 68, cin), np.float32)
    xp[1:H + 1, 1:W + 1] = x_hwc                                  # patch origin (-1, -1)
    f = np.zeros((H, W, cp), np.float32)
    m = np.zeros((H, W, cp), np.float32)
    t, kl = LANE & 15, LANE >> 4
    slot = kl ^ ((t >> 1) & 3)
    tid = np.arange(256)
    tc16, tt16 = tid & 15, tid >> 4                               # transform role: thread = (input channel of the chunk, tile)
    for by in range((H + 7) // 8):
        for bx in range((W + 31) // 32):
            oy0, ox0 = 8 * by, 32 * bx
            for g in range(cp // 32):
                acc = np.zeros((4, 36, 64, 4), np.float32)                    # [wave][frequency][lane][register]
                for c in range(cin // 16):
                    vbuf = np.full((36, 16, 16), np.nan, np.float32)          # [frequency][tile][swizzled channel]
                    for th in range(256):                                     # one (tile, channel) 6x6 patch per thread
                        tl, cc = tt16[th], tc16[th]
                        tr, tcol = tl >> 3, tl & 7
                        d = xp[oy0 + 4 * tr:oy0 + 4 * tr + 6, ox0 + 4 * tcol:ox0 + 4 * tcol + 6, 16 * c + cc]
                        V = BT @ d @ BT.T
                        sw = ((cc >> 2) ^ ((tl >> 1) & 3)) * 4 + (cc & 3)
                        vbuf[:, tl, sw] = V.reshape(36)
                    assert not np.isnan(vbuf).any()
                    for w in range(4):
                        for fq in range(36):
                            Bop = vbuf[fq][t][:, None, :].reshape(64, 16)[np.arange(64)[:, None], (slot * 4)[:, None] + np.arange(4)[None, :]]
                            for e in range(4):
                                acc[w, fq] += mfma_16x16x4(P[g, w, c, fq][:, e], Bop[:, e])
                q = LANE >> 4
                for w in range(4):
                    for r in range(4):
                        Mx = acc[w, :, :, r].reshape(6, 6, 64)
                        Y = np.einsum("pa,ajl,qj->pql", AT, Mx, AT)           # (4,4,64)
                        i_row = 4 * q + r
                        ch = 32 * g + 8 * w + (i_row & 7)
                        for l in range(64):
                            tr_l, tc_l = (l & 15) >> 3, (l & 15) & 7
                            for py in range(4):
                                for px in range(4):
                                    oy, ox = oy0 + 4 * tr_l + py, ox0 + 4 * tc_l + px
                                    if oy < H and ox < W:
                                        (m if i_row[l] >= 8 else f)[oy, ox, ch[l]] = Y[py, px, l]
    return f[:, :, :cout], m[:, :, :cout]
