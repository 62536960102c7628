"""NumPy model of the wave-autonomous Winograd kernels on v_mfma_f32_16x16x4_f32 (gated_conv_wino16_kernel in
read_amd/csrc/conv.hip), written with the SAME index maps as the HIP kernel — filter transform + fragment packing, per-lane
input transform, the 16x16x4 MFMA operand / result lane maps, the in-lane output transform and the pixel / channel each
lane finishes — so that layout mistakes show up on the CPU (tests/test_wino_cpu.py compares with torch's conv2d).

F(2x2,3x3):  a workgroup unit = 4 x 8 tiles (8 x 16 output pixels) x 32 output channels; wave w owns channels 8w .. 8w+7 of the
group, conv_f in MFMA rows 0..7 and conv_m in rows 8..15 (A operand = weights), and BOTH 16-tile halves of the block (B operand =
transformed input of tiles (2b + t>>3, t&7), b = 0, 1): 16 frequencies x 2 blocks x 4 registers = 128 accumulators per lane,
every frequency of a (tile, channel) pair in the same lane -> the output transform needs no other wave.
"""
import numpy as np

BT = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], np.float32)
G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], np.float32)
AT = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], np.float32)
LANE = np.arange(64)


def filter_transform(w):
    """w (Cout,Cin,3,3) -> U (4,4,Cin,Cout)."""
    return np.einsum("ia,ocab,jb->ijco", G, w.astype(np.float32), G).astype(np.float32)


def pack_w16(wf, wm):
    """-> flat fp32 [group][wave 4][chunk of 16 cin][a 4][j 4][lane 64][e 4]:
    lane (i = lane & 15, kl = lane >> 4) holds U_{f if i < 8 else m}[a][j][cin = 16 chunk + 4 kl + e][cout = 32 g + 8 w + (i & 7)]."""
    cout, cin = wf.shape[:2]
    cp = (cout + 31) // 32 * 32
    U = [filter_transform(wf), filter_transform(wm)]
    out = np.zeros((cp // 32, 4, cin // 16, 4, 4, 64, 4), np.float32)
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
                        for a in range(4):
                            for j in range(4):
                                out[g, w, c, a, j, sel, e] = U[fm][a, j, ci[sel], co[sel]]
    return out.reshape(-1)


def mfma_16x16x4(a_lane, b_lane):
    """One v_mfma_f32_16x16x4_f32 with C = 0: a_lane / b_lane (64,) per-lane operand registers -> D as (64, 4) result registers.
    A[i = l & 15][k = l >> 4], B[k = l >> 4][j = l & 15], D[i = 4 (l >> 4) + r][j = l & 15]."""
    A = np.zeros((16, 4), np.float32)
    B = np.zeros((4, 16), np.float32)
    A[LANE & 15, LANE >> 4] = a_lane
    B[LANE >> 4, LANE & 15] = b_lane
    D = A @ B
    out = np.zeros((64, 4), np.float32)
    for r in range(4):
        out[:, r] = D[4 * (LANE >> 4) + r, LANE & 15]
    return out


def wino16_conv_model(x_hwc, packed, cin, cout):
    """x (H,W,Cin) NHWC -> (f, m) pre-activation maps (H,W,Cout), through the kernel's lane maps."""
    H, W, _ = x_hwc.shape
    cp = (cout + 31) // 32 * 32
    P = packed.reshape(cp // 32, 4, cin // 16, 4, 4, 64, 4)
    xp = np.zeros((H + 18, W + 34, cin), np.float32)
    xp[1:H + 1, 1:W + 1] = x_hwc                                  # patch origin (-1, -1): input row oy - 1 at index oy
    f = np.zeros((H, W, cp), np.float32)
    m = np.zeros((H, W, cp), np.float32)
    t, kl = LANE & 15, LANE >> 4
    trp, tc = t >> 3, t & 7
    rows = {0: (0, 2, 1.0, -1.0), 1: (1, 2, 1.0, 1.0), 2: (2, 1, 1.0, -1.0), 3: (1, 3, 1.0, -1.0)}     # T[a] = sa d[ra] + sb d[rb]
    for by in range((H + 7) // 8):
        for bx in range((W + 15) // 16):
            oy0, ox0 = 8 * by, 16 * bx
            for g in range(cp // 32):
                for w in range(4):
                    acc = np.zeros((2, 4, 4, 64, 4), np.float32)              # [block][a][j][lane][register]
                    for c in range(cin // 16):
                        for a in range(4):
                            ra, rb, sa, sb = rows[a]
                            for b in range(2):
                                tr = 2 * b + trp
                                T = np.zeros((4, 64, 4), np.float32)
                                for cc in range(4):
                                    for e in range(4):
                                        ci = 16 * c + 4 * kl + e
                                        T[cc, :, e] = sa * xp[oy0 + 2 * tr + ra, ox0 + 2 * tc + cc, ci] + \
                                                      sb * xp[oy0 + 2 * tr + rb, ox0 + 2 * tc + cc, ci]
                                V = [T[0] - T[2], T[1] + T[2], T[2] - T[1], T[1] - T[3]]
                                for e in range(4):                            # k-step e: cin 16 c + 4 kl + e
                                    for j in range(4):
                                        acc[b, a, j] += mfma_16x16x4(P[g, w, c, a, j][:, e], V[j][:, e])
                    # ---- in-lane output transform + where each lane's results go
                    q = LANE >> 4
                    for b in range(2):
                        for r in range(4):
                            M = acc[b, :, :, :, r]                            # (4,4,64): [a][j][lane]
                            Y = np.einsum("pa,ajl,qj->pql", AT, M, AT)        # (2,2,64)
                            i_row = 4 * q + r                                 # MFMA row = channel slot
                            ch = 32 * g + 8 * w + (i_row & 7)
                            is_m = i_row >= 8
                            for l in range(64):
                                tr_l, tc_l = 2 * b + (l & 15) // 8, (l & 15) % 8
                                for pa in range(2):
                                    for pb in range(2):
                                        oy, ox = oy0 + 2 * tr_l + pa, ox0 + 2 * tc_l + pb
                                        if oy < H and ox < W:
                                            (m if is_m[l] else f)[oy, ox, ch[l]] = Y[pa, pb, l]
    return f[:, :, :cout], m[:, :, :cout]


# ---- version 2 of the kernel: the input transform is SHARED by the four waves through LDS -------------------------------------
# (the fp32 MFMA runs on the same FP pipe as the VALU — tools/issue_probe.py — so transform instructions are not hidden
# behind MFMAs; version 1 let every wave transform all 32 tiles for itself, 4x the necessary VALU work)
#   part p of a chunk = frequency rows a = 2p, 2p + 1;  transform role of wave w: block tb = w >> 1, row a = 2p + (w & 1);
#   lane (t = lane & 15, quad = lane >> 4) forms V[a][0..3] of tile 16 tb + t for input channels 4 quad .. 4 quad + 3 and stores
#   them at  Vbuf[p][fl = 4 (w & 1) + j][tile][slot (quad ^ ((t >> 1) & 3))]   (float4 slots; the XOR keeps both the stores and
#   the MFMA-side ds_read_b128 of lanes (t, kl) bank-conflict free);  MFMA role as before, B operand = Vbuf[p][fl][16 b + t][slot].
def wino16v2_conv_model(x_hwc, packed, cin, cout):
    H, W, _ = x_hwc.shape
    cp = (cout + 31) // 32 * 32
    P = packed.reshape(cp // 32, 4, cin // 16, 4, 4, 64, 4)
    xp = np.zeros((H + 18, W + 34, cin), np.float32)
    xp[1:H + 1, 1:W + 1] = x_hwc
    f = np.zeros((H, W, cp), np.float32)
    m = np.zeros((H, W, cp), np.float32)
    t, kl = LANE & 15, LANE >> 4
    slot = kl ^ ((t >> 1) & 3)
    rows = {0: (0, 2, 1.0, -1.0), 1: (1, 2, 1.0, 1.0), 2: (2, 1, 1.0, -1.0), 3: (1, 3, 1.0, -1.0)}
    for by in range((H + 7) // 8):
        for bx in range((W + 15) // 16):
            oy0, ox0 = 8 * by, 16 * bx
            for g in range(cp // 32):
                acc = np.zeros((4, 2, 4, 4, 64, 4), np.float32)              # [wave][block][a][j][lane][register]
                for c in range(cin // 16):
                    for part in range(2):
                        vbuf = np.full((8, 32, 4, 4), np.nan, np.float32)     # [fl][tile][slot][e]
                        for w in range(4):                                    # transform role
                            tb, a = w >> 1, 2 * part + (w & 1)
                            ra, rb, sa, sb = rows[a]
                            tr, tcc = 2 * tb + (t >> 3), t & 7
                            T = np.zeros((4, 64, 4), np.float32)
                            for cc in range(4):
                                for e in range(4):
                                    ci = 16 * c + 4 * kl + e
                                    T[cc, :, e] = sa * xp[oy0 + 2 * tr + ra, ox0 + 2 * tcc + cc, ci] + \
                                                  sb * xp[oy0 + 2 * tr + rb, ox0 + 2 * tcc + cc, ci]
                            V = [T[0] - T[2], T[1] + T[2], T[2] - T[1], T[1] - T[3]]
                            for j in range(4):
                                vbuf[4 * (w & 1) + j, 16 * tb + t, slot] = V[j]
                        assert not np.isnan(vbuf).any()                       # every slot written exactly by someone
                        for w in range(4):                                    # MFMA role
                            for fl in range(8):
                                a, j = 2 * part + (fl >> 2), fl & 3
                                for b in range(2):
                                    Bop = vbuf[fl, 16 * b + t, slot]          # (64, 4)
                                    for e in range(4):
                                        acc[w, b, a, j] += mfma_16x16x4(P[g, w, c, a, j][:, e], Bop[:, e])
                q = LANE >> 4
                for w in range(4):
                    for b in range(2):
                        for r in range(4):
                            Y = np.einsum("pa,ajl,qj->pql", AT, acc[w, b, :, :, :, r], AT)
                            i_row = 4 * q + r
                            ch = 32 * g + 8 * w + (i_row & 7)
                            for l in range(64):
                                tr_l, tc_l = 2 * b + (l & 15) // 8, (l & 15) % 8
                                for pa in range(2):
                                    for pb in range(2):
                                        oy, ox = oy0 + 2 * tr_l + pa, ox0 + 2 * tc_l + pb
                                        if oy < H and ox < W:
                                            (m if i_row[l] >= 8 else f)[oy, ox, ch[l]] = Y[pa, pb, l]
    return f[:, :, :cout], m[:, :, :cout]
