"""NumPy restatement of the direct split-operand 3x3 kernel's operand and arithmetic (gated_conv_d3h_kernel in read_amd/csrc/conv.hip):
the host packer (row scales, f16 hi / lo pieces, fragment order) and the product it stands for — every tap executed,
    w x  ~=  [ (2^-11 wh) xl + wl xh + wh xh ] / s,    x = xh + 2^-11 xl,   w s = wh + wl   (all four f16, round to nearest even).
tests/test_wino_cpu.py compares the library's packer with this one bit for bit and the arithmetic with torch's conv2d."""
import numpy as np

LANE = np.arange(64)


def row_scale_exp(w):
    """w (Cout, Cin, 3, 3) -> per output channel the exponent ex with max |w| 2^ex in [2^14, 2^15) (0 for an all-zero row)."""
    mx = np.abs(w.astype(np.float64)).max(axis=(1, 2, 3))
    ex = np.zeros(mx.shape, np.int64)
    nz = mx > 0
    _, e = np.frexp(mx[nz])
    ex[nz] = np.clip(15 - e, -60, 60)
    return ex


def pack_d3h(wf, wm):
    """-> (halfs float16 [group][row half 2][chunk of 32][tap 9][row block 2][piece 2][lane 64][8], inv float32 [2][CoutPad]); lane
    (i = lane & 15, kq = lane >> 4) holds row i of the block (i < 8: conv_f of channel 32 g + 16 rh + 8 rb + i, else conv_m of channel
    ... + i - 8), input channels 32 chunk + 8 kq + e."""
    cout, cin = wf.shape[:2]
    cp = (cout + 31) // 32 * 32
    halfs = np.zeros((cp // 32, 2, cin // 32, 9, 2, 2, 64, 8), np.float16)
    inv = np.ones((2, cp), np.float32)
    i, kq = LANE & 15, LANE >> 4
    for fm, w_ in enumerate((wf, wm)):
        ex = row_scale_exp(w_)
        ws = np.ldexp(w_.astype(np.float64), ex[:, None, None, None])
        hi = ws.astype(np.float16)
        lo = (ws - hi.astype(np.float64)).astype(np.float16)
        inv[fm, :cout] = np.ldexp(1.0, -ex).astype(np.float32)
        for g in range(cp // 32):
            for rh in range(2):
                for rb in range(2):
                    co = 32 * g + 16 * rh + 8 * rb + (i & 7)
                    sel = (co < cout) & ((i >> 3) == fm)
                    for c in range(cin // 32):
                        for e in range(8):
                            ci = 32 * c + 8 * kq + e
                            for tap in range(9):
                                halfs[g, rh, c, tap, rb, 0, sel, e] = hi[co[sel], ci[sel], tap // 3, tap % 3]
                                halfs[g, rh, c, tap, rb, 1, sel, e] = lo[co[sel], ci[sel], tap // 3, tap % 3]
    return halfs, inv


def pack_d3h_blob(wf, wm):
    halfs, inv = pack_d3h(wf, wm)
    return np.concatenate([halfs.reshape(-1).view(np.float32), inv.reshape(-1)])


def split_conv_model(x_hwc, wf):
    """The kernel's arithmetic for ONE of the two convolutions, dense (no lane maps): three piece pairs, products exact, sums in float64
    (the device keeps at least that per 32-channel block), -> (H, W, Cout) fp32."""
    H, W, cin = x_hwc.shape
    cout = wf.shape[0]
    ex = row_scale_exp(wf)
    ws = np.ldexp(wf.astype(np.float64), ex[:, None, None, None])
    wh = ws.astype(np.float16)
    wl = (ws - wh.astype(np.float64)).astype(np.float16).astype(np.float64)
    whs = (wh * np.float16(2.0 ** -11)).astype(np.float16).astype(np.float64)
    wh = wh.astype(np.float64)
    xp = np.zeros((H + 2, W + 2, cin), np.float32)
    xp[1:-1, 1:-1] = x_hwc
    xh = xp.astype(np.float16)
    xl = ((xp - xh.astype(np.float32)) * np.float32(2048.0)).astype(np.float16).astype(np.float64)
    xh = xh.astype(np.float64)
    out = np.zeros((H, W, cout), np.float64)
    for ky in range(3):
        for kx in range(3):
            a, b = xh[ky:ky + H, kx:kx + W], xl[ky:ky + H, kx:kx + W]
            out += b @ whs[:, :, ky, kx].T + a @ wl[:, :, ky, kx].T + a @ wh[:, :, ky, kx].T
    return (out * np.ldexp(1.0, -ex)[None, None, :]).astype(np.float32)


def pack_d1h(wf, wm):
    """The 1x1 layers' operand (gated_conv_pxh_kernel, read_conv_pack_dkh_host with ksize 1): wf, wm (Cout, Cin) ->
    (halfs float16 [k16 step][tile = 2 group + (f | m)][piece 2][lane 64][8], inv float32 [2][CoutPad]); lane (i = lane & 31, h = lane >> 5)
    holds row i of the tile (conv_f / conv_m of channel 32 group + i), input channels 16 step + 8 h + e.  Scales and pieces as pack_d3h."""
    cout, cin = wf.shape[:2]
    cp = (cout + 31) // 32 * 32
    halfs = np.zeros((cin // 16, cp // 16, 2, 64, 8), np.float16)
    inv = np.ones((2, cp), np.float32)
    i, h = LANE & 31, LANE >> 5
    for fm, w_ in enumerate((wf, wm)):
        w4 = w_.reshape(cout, cin, 1, 1)
        ex = row_scale_exp(w4)
        ws = np.ldexp(w4.astype(np.float64), ex[:, None, None, None])[:, :, 0, 0]
        hi = ws.astype(np.float16)
        lo = (ws - hi.astype(np.float64)).astype(np.float16)
        inv[fm, :cout] = np.ldexp(1.0, -ex).astype(np.float32)
        for g in range(cp // 32):
            co = 32 * g + i
            sel = co < cout
            for st in range(cin // 16):
                for e in range(8):
                    ci = 16 * st + 8 * h + e
                    halfs[st, 2 * g + fm, 0, sel, e] = hi[co[sel], ci[sel]]
                    halfs[st, 2 * g + fm, 1, sel, e] = lo[co[sel], ci[sel]]
    return halfs, inv


def pack_d1h_blob(wf, wm):
    halfs, inv = pack_d1h(np.asarray(wf, np.float32).reshape(wf.shape[0], -1), np.asarray(wm, np.float32).reshape(wm.shape[0], -1))
    return np.concatenate([halfs.reshape(-1).view(np.float32), inv.reshape(-1)])


def split_1x1_model(x_hwc, wf):
    """The pixel-lane kernel's arithmetic for one of the two 1x1 convolutions, dense: -> (H, W, Cout) fp32."""
    return split_conv_model_taps(x_hwc, np.asarray(wf, np.float32).reshape(wf.shape[0], -1))


def split_conv_model_taps(x_hwc, w2):
    cout, cin = w2.shape
    ex = row_scale_exp(w2.reshape(cout, cin, 1, 1))
    ws = np.ldexp(w2.astype(np.float64), ex[:, None])
    wh = ws.astype(np.float16)
    wl = (ws - wh.astype(np.float64)).astype(np.float16).astype(np.float64)
    whs = (wh * np.float16(2.0 ** -11)).astype(np.float16).astype(np.float64)
    wh = wh.astype(np.float64)
    xh = x_hwc.astype(np.float16)
    xl = ((x_hwc - xh.astype(np.float32)) * np.float32(2048.0)).astype(np.float16).astype(np.float64)
    xh = xh.astype(np.float64)
    out = xl @ whs.T + xh @ wl.T + xh @ wh.T
    return (out * np.ldexp(1.0, -ex)[None, None, :]).astype(np.float32)
