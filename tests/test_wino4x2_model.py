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
