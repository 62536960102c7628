"""CPU: the algebra of the two-waves-per-SIMD split of the F(4x4,3x3) kernel sketched in DESIGN.md 12.1(d) — NOT a kernel in the
library yet; the model pins the pieces a kernel would be built from, so that they are known to be right before any HIP is written.

  * frequencies split over a wave pair: half h owns frequency rows 3h .. 3h+2 of the 6 x 6 grid (18 of the 36 accumulators);
  * input transform by halves: thread (channel, tile, h) reads the whole 6 x 6 patch and forms rows 3h .. 3h+2 of B^T d, then
    their products with B — exactly its half's 18 frequencies;
  * output transform by halves: the column pass T[xi][q] = sum_nu M[xi][nu] A^T[q][nu] is local to a frequency row; the row pass is
    linear in the rows, so half h forms partial sums P_h[p][q] of all four output rows from its three rows with the coefficient
    patterns below, keeps rows 2h, 2h+1 and hands the other two to its partner: Y = P_0 + P_1.
"""
import numpy as np

from tests.wino4_ref import AT, BT


def _col_pass(M_row):
    """one frequency row (6 values per output column position) -> T[0..3]: the kernel's 10-operation form"""
    s1, d1, s2, d2 = M_row[1] + M_row[2], M_row[1] - M_row[2], M_row[3] + M_row[4], M_row[3] - M_row[4]
    return np.array([M_row[0] + s1 + s2, d1 + 2 * d2, s1 + 4 * s2, d1 + 8 * d2 + M_row[5]])


def test_output_transform_splits_over_frequency_rows():
    rng = np.random.default_rng(5)
    M = rng.standard_normal((6, 6)).astype(np.float64)                  # accumulators of one (tile, channel): M[xi][nu]
    want = AT.astype(np.float64) @ M @ AT.astype(np.float64).T          # Y = A^T M A
    T = np.stack([_col_pass(M[xi]) for xi in range(6)])                 # (6 frequency rows, 4 output columns)
    # half 0 (rows 0, 1, 2):  P0 = T0 + s, P1 = d, P2 = s, P3 = d        with s = T1 + T2, d = T1 - T2            (3 operations per column)
    s, d = T[1] + T[2], T[1] - T[2]
    P_0 = np.stack([T[0] + s, d, s, d])
    # half 1 (rows 3, 4, 5):  P0 = s, P1 = 2 d, P2 = 4 s, P3 = 8 d + T5  with s = T3 + T4, d = T3 - T4            (5 operations per column)
    s, d = T[3] + T[4], T[3] - T[4]
    P_1 = np.stack([s, 2 * d, 4 * s, 8 * d + T[5]])
    assert np.abs(P_0 + P_1 - want).max() <= 1e-12
    # what crosses the wave pair: half h keeps output rows 2h, 2h + 1 and receives the partner's partials of those rows
    for h, (own, other) in enumerate(((P_0, P_1), (P_1, P_0))):
        rows = slice(2 * h, 2 * h + 2)
        assert np.abs(own[rows] + other[rows] - want[rows]).max() <= 1e-12


def test_input_transform_by_frequency_row_halves():
    rng = np.random.default_rng(6)
    d = rng.standard_normal((6, 6)).astype(np.float64)
    B = BT.astype(np.float64)
    V = B @ d @ B.T
    for h in range(2):
        rows = slice(3 * h, 3 * h + 3)
        first = B[rows] @ d                                              # three rows of B^T d: needs the whole patch
        assert np.abs(first @ B.T - V[rows]).max() <= 1e-12              # ... and gives exactly the half's 18 frequencies (6 xi + nu)
    # the per-column forms a kernel would use for the first pass (rows 0..2 and rows 3..5 of B^T applied to a column d0..d5)
    c = d[:, 0]
    a, b, cc, e = c[4] - 4 * c[2], c[3] - 4 * c[1], c[4] - c[2], c[3] - c[1]
    top = np.array([4 * c[0] - 5 * c[2] + c[4], a + b, a - b])
    bot = np.array([cc + 2 * e, cc - 2 * e, 4 * c[1] - 5 * c[3] + c[5]])
    assert np.abs(np.concatenate([top, bot]) - B @ c).max() <= 1e-12


def test_lane_level_model_of_the_split_kernel():
    """The index maps of gated_conv_wino4x2_kernel (csrc/conv.hip), mirrored expression by expression: thread (c16, tile, h) of the
    half transforms and the V slots it writes, wave (co, fh) with its half of the octet's weight fragments, the 18 accumulators,
    the partial row pass with keep / give, the hand-over between waves w and w ^ 4, the half-wave swap that pairs conv_f with
    conv_m, and the pixel / channel each lane ends up storing — against torch's conv2d for both branches."""
    import torch
    import torch.nn.functional as F
    from tests.wino4_ref import LANE, mfma_16x16x4, pack_w4
    rng = np.random.default_rng(11)
    cin, cout, H, W = 32, 32, 8, 32                                     # one unit, two 16-channel chunks, one channel group
    wf = rng.standard_normal((cout, cin, 3, 3)).astype(np.float32) * 0.2
    wm = rng.standard_normal((cout, cin, 3, 3)).astype(np.float32) * 0.2
    x = rng.standard_normal((H, W, cin)).astype(np.float32)
    P = pack_w4(wf, wm).reshape(1, 4, cin // 16, 36, 64, 4)
    xp = np.zeros((H + 20, W + 68, cin), np.float32)
    xp[1:H + 1, 1:W + 1] = x
    t16, kl = LANE & 15, LANE >> 4
    slot = kl ^ ((t16 >> 1) & 3)
    acc = np.zeros((8, 18, 64, 4), np.float32)                          # [wave][local frequency][lane][register]
    tid = np.arange(512)
    for c in range(cin // 16):
        vbuf = np.full((36, 16, 16), np.nan, np.float32)
        for th in tid:                                                  # transform role: c16 = tid & 15, tl = (tid >> 4) & 15, h = tid >> 8
            c16, tl, h = th & 15, (th >> 4) & 15, th >> 8
            d = xp[4 * (tl >> 3):4 * (tl >> 3) + 6, 4 * (tl & 7):4 * (tl & 7) + 6, 16 * c + c16]
            rows = BT[3 * h:3 * h + 3] @ d                              # bt3v: rows 3 h .. 3 h + 2 of B^T d, all six columns
            V = rows @ BT.T                                             # bt6row on each of the three rows
            sw = ((c16 >> 2) ^ ((tl >> 1) & 3)) * 4 + (c16 & 3)
            for r in range(3):
                for nu in range(6):
                    vbuf[(3 * h + r) * 6 + nu, tl, sw] = V[r, nu]
        assert not np.isnan(vbuf).any()
        for w in range(8):
            co, fh = w & 3, w >> 2
            for fql in range(18):
                fq = 18 * fh + fql
                Bop = vbuf[fq][t16][:, None, :].reshape(64, 16)[np.arange(64)[:, None], (slot * 4)[:, None] + np.arange(4)[None, :]]
                for e in range(4):
                    acc[w, fql] += mfma_16x16x4(P[0, co, c, fq][:, e], Bop[:, e])
    # epilogue per wave: column pass, partial row pass
    keep = np.zeros((8, 2, 4, 64, 4), np.float32)
    give = np.zeros((8, 2, 4, 64, 4), np.float32)
    for w in range(8):
        fh = w >> 2
        T = np.zeros((3, 4, 64, 4), np.float32)
        for r in range(3):
            M = acc[w, 6 * r:6 * r + 6]
            s1, d1, s2, dd = M[1] + M[2], M[1] - M[2], M[3] + M[4], M[3] - M[4]
            T[r] = np.stack([M[0] + s1 + s2, d1 + 2 * dd, s1 + 4 * s2, d1 + 8 * dd + M[5]])
        for q in range(4):
            if fh == 0:
                sm, df = T[1][q] + T[2][q], T[1][q] - T[2][q]
                keep[w, 0, q], keep[w, 1, q], give[w, 0, q], give[w, 1, q] = T[0][q] + sm, df, sm, df
            else:
                sm, df = T[0][q] + T[1][q], T[0][q] - T[1][q]
                give[w, 0, q], give[w, 1, q], keep[w, 0, q], keep[w, 1, q] = sm, 2 * df, 4 * sm, 8 * df + T[2][q]
    f = np.zeros((H, W, cout), np.float32)
    m = np.zeros((H, W, cout), np.float32)
    for w in range(8):
        co, fh = w & 3, w >> 2
        Y = keep[w] + give[w ^ 4]                                       # the partner's message: its `give` rows are this wave's rows, same (o, q) slots
        for px in range(4):
            u0, u1 = Y[0, px].copy(), Y[1, px].copy()                   # rows 2 fh and 2 fh + 1; lanes 0..31 conv_f, 32..63 conv_m
            u0n, u1n = u0.copy(), u1.copy()
            u0n[32:], u1n[:32] = u1[:32], u0[32:]                       # v_permlane32_swap: upper half of u0 <-> lower half of u1
            for l in range(64):
                cq, hf = (l >> 4) & 1, l >> 5
                c0 = co * 8 + 4 * cq
                oy = 4 * ((l & 15) >> 3) + 2 * fh + hf
                ox = 4 * ((l & 15) & 7) + px
                f[oy, ox, c0:c0 + 4] = u0n[l]
                m[oy, ox, c0:c0 + 4] = u1n[l]
    xt = torch.from_numpy(x).permute(2, 0, 1)[None]
    ref_f = F.conv2d(xt, torch.from_numpy(wf), padding=1)[0].permute(1, 2, 0).numpy()
    ref_m = F.conv2d(xt, torch.from_numpy(wm), padding=1)[0].permute(1, 2, 0).numpy()
    assert np.abs(f - ref_f).max() <= 2e-4 * (1 + np.abs(ref_f).max()), np.abs(f - ref_f).max()
    assert np.abs(m - ref_m).max() <= 2e-4 * (1 + np.abs(ref_m).max()), np.abs(m - ref_m).max()
