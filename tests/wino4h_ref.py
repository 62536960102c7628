"""NumPy model of the split-operand Winograd F(4x4,3x3) kernel on v_mfma_f32_16x16x32_f16 (gated_conv_wino4h_kernel in
read_amd/csrc/conv.hip): the host packer (row scales, f16 hi / lo pieces, fragment order), the transform thread's (tile, channel
pair) role, the hi / scaled-lo split of the transformed input, the swizzled V buffer, the MFMA operand and result lane maps, the
three piece pairs per product and the 1 / s of the epilogue.  tests/test_wino_cpu.py compares it with torch's conv2d and the
library's packer with this one, bit for bit.

Arithmetic (DESIGN.md 3.3 a+):  V = Vh + 2^-11 Vl,  U s = Uh + Ul  (all four f16, round to nearest even),
    U V  ~=  [ (2^-11 Uh) Vl + Ul Vh + Uh Vh ] / s        three MFMAs into one fp32 accumulator per 32 input channels.
"""
import numpy as np

from tests.wino4_ref import AT, BT, G, LANE


def filter_transform4_f64(w):
    return np.einsum("ia,ocab,jb->ijco", G, w.astype(np.float64), G)                # (6,6,Cin,Cout), float64


def row_scale_exp(U):
    """U (6,6,Cin,Cout) float64 -> per output channel the exponent ex with max |U| 2^ex in [2^14, 2^15) (0 for an all-zero row)."""
    mx = np.abs(U).max(axis=(0, 1, 2))
    ex = np.zeros(mx.shape, np.int64)
    nz = mx > 0
    _, e = np.frexp(mx[nz])                                                         # mx in [2^(e-1), 2^e)
    ex[nz] = np.clip(15 - e, -60, 60)
    return ex


def pack_w4h(wf, wm):
    """-> (halfs uint16 [group][wave 4][chunk of 32][frequency 36][piece 2][lane 64][8], inv float32 [2][CoutPad]); lane
    (i = lane & 15, kq = lane >> 4) holds row i (conv_f of channel 32 g + 8 w + i for i < 8, conv_m of channel ... + i - 8 else),
    input channels 32 chunk + 8 kq + e."""
    cout, cin = wf.shape[:2]
    cp = (cout + 31) // 32 * 32
    halfs = np.zeros((cp // 32, 4, cin // 32, 36, 2, 64, 8), np.float16)
    inv = np.ones((2, cp), np.float32)
    i, kq = LANE & 15, LANE >> 4
    for fm, w_ in enumerate((wf, wm)):
        U = filter_transform4_f64(w_)
        ex = row_scale_exp(U)
        Us = np.ldexp(U, ex[None, None, None, :])
        hi = Us.astype(np.float16)
        lo = (Us - hi.astype(np.float64)).astype(np.float16)
        inv[fm, :cout] = np.ldexp(1.0, -ex).astype(np.float32)
        for g in range(cp // 32):
            for w in range(4):
                co = 32 * g + 8 * w + (i & 7)
                sel = (co < cout) & ((i >> 3) == fm)
                for c in range(cin // 32):
                    for e in range(8):
                        ci = 32 * c + 8 * kq + e
                        for fq in range(36):
                            halfs[g, w, c, fq, 0, sel, e] = hi[fq // 6, fq % 6, ci[sel], co[sel]]
                            halfs[g, w, c, fq, 1, sel, e] = lo[fq // 6, fq % 6, ci[sel], co[sel]]
    return halfs, inv


def pack_w4h_blob(wf, wm):
    """The packer's output as the library lays it out: float32 words, halfs first, then 1 / s."""
    halfs, inv = pack_w4h(wf, wm)
    return np.concatenate([halfs.reshape(-1).view(np.float32), inv.reshape(-1)])


def mfma_16x16x32(a_lane, b_lane):
    """a_lane, b_lane (64, 8) f16 -> (64, 4) fp32: lane l holds A[l & 15][8 (l >> 4) + e] / B[8 (l >> 4) + e][l & 15];
    D row 4 (l >> 4) + r, column l & 15.  (exact products, summed here in float64 and rounded once: the device keeps at least
    that much, profiles/r6_f16split_probe.txt)"""
    A = np.zeros((16, 32), np.float64)
    B = np.zeros((32, 16), np.float64)
    for e in range(8):
        A[LANE & 15, 8 * (LANE >> 4) + e] = a_lane[:, e]
        B[8 * (LANE >> 4) + e, LANE & 15] = b_lane[:, e]
    D = A @ B
    out = np.zeros((64, 4), np.float64)
    for r in range(4):
        out[:, r] = D[4 * (LANE >> 4) + r, LANE & 15]
    return out


def wino4h_conv_model(x_hwc, halfs, inv, cin, cout):
    H, W, _ = x_hwc.shape
    cp = (cout + 31) // 32 * 32
    xp = np.zeros((H + 20, W + 68, cin), np.float32)
    xp[1:H + 1, 1:W + 1] = x_hwc
    f = np.zeros((H, W, cp), np.float32)
    m = np.zeros((H, W, cp), np.float32)
    t, kl = LANE & 15, LANE >> 4
    rd_slot = kl ^ ((-(t >> 2)) & 3)                                               # B operand: 16-byte slot of (tile t, k octet kl)
    k11 = np.float16(2.0 ** -11)
    for by in range((H + 7) // 8):
        for bx in range((W + 31) // 32):
            oy0, ox0 = 8 * by, 32 * bx
            for g in range(cp // 32):
                acc = np.zeros((4, 36, 64, 4), np.float32)
                for c in range(cin // 32):
                    vbuf = np.full((36, 2, 16, 4, 8), np.nan, np.float16)           # [frequency][piece][tile][slot][8 halfs]
                    for th in range(256):                                           # thread = (tile, channel pair)
                        wv, lane = th >> 6, th & 63
                        cpair, tl = lane & 15, wv * 4 + (lane >> 4)
                        tr, tcol = tl >> 3, tl & 7
                        d = xp[oy0 + 4 * tr:oy0 + 4 * tr + 6, ox0 + 4 * tcol:ox0 + 4 * tcol + 6, 32 * c + 2 * cpair:32 * c + 2 * cpair + 2]
                        V = np.einsum("ia,abk,jb->ijk", BT, d, BT).astype(np.float32).reshape(36, 2)
                        hi = V.astype(np.float16)
                        lo = ((V - hi.astype(np.float32)) * np.float32(2048.0)).astype(np.float16)
                        slot = (cpair >> 2) ^ ((-wv) & 3)
                        e0 = 2 * (cpair & 3)
                        vbuf[:, 0, tl, slot, e0:e0 + 2] = hi
                        vbuf[:, 1, tl, slot, e0:e0 + 2] = lo
                    assert not np.isnan(vbuf.astype(np.float32)).any()
                    for w in range(4):
                        for fq in range(36):
                            bh, bl = vbuf[fq, 0, t, rd_slot], vbuf[fq, 1, t, rd_slot]             # (64, 8)
                            ah, al = halfs[g, w, c, fq, 0], halfs[g, w, c, fq, 1]
                            ahs = (ah * k11).astype(np.float16)
                            s = mfma_16x16x32(ahs, bl) + mfma_16x16x32(al, bh) + mfma_16x16x32(ah, bh)
                            acc[w, fq] = (acc[w, fq].astype(np.float64) + s).astype(np.float32)
                q = LANE >> 4
                for w in range(4):
                    for r in range(4):
                        Mx = acc[w, :, :, r].reshape(6, 6, 64)
                        Y = np.einsum("pa,ajl,qj->pql", AT, Mx, AT)
                        i_row = 4 * q + r
                        ch = 32 * g + 8 * w + (i_row & 7)
                        for l in range(64):
                            tr_l, tc_l = (l & 15) >> 3, (l & 15) & 7
                            isc = inv[1 if i_row[l] >= 8 else 0, ch[l]]
                            for py in range(4):
                                for px in range(4):
                                    oy, ox = oy0 + 4 * tr_l + py, ox0 + 4 * tc_l + px
                                    if oy < H and ox < W:
                                        (m if i_row[l] >= 8 else f)[oy, ox, ch[l]] = Y[py, px, l] * isc
    return f[:, :, :cout], m[:, :, :cout]
