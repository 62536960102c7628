"""CPU: the numerics claim behind DESIGN.md 12.1 (c), pinned — Winograd F(4x4,3x3) products formed from exact bf16 pieces of the fp32
operands (tools/bf16x3_study.py).  No kernel is involved: this is the arithmetic a split-operand kernel would have to reproduce."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import bf16x3_study as st                                                            # noqa: E402


def test_three_bf16_pieces_carry_an_fp32_exactly():
    rng = np.random.default_rng(1)
    x = (rng.standard_normal(100_000) * np.exp(rng.uniform(-20, 20, 100_000))).astype(np.float32)
    h, m, l = st.split3(x)
    for p in (h, m, l):
        assert not np.any(p.view(np.uint32) & 0xFFFF), "a piece is not a bf16"
    assert np.array_equal((h.astype(np.float64) + m.astype(np.float64) + l.astype(np.float64)).astype(np.float32), x)
    # round to nearest even on the cut: 1 + 2^-8 lies half way between two bf16 neighbours and goes to the even one
    assert st.bf16_round(np.float32(1.0 + 2.0 ** -8)) == np.float32(1.0)
    assert st.bf16_round(np.float32(1.0 + 3 * 2.0 ** -8)) == np.float32(1.0 + 2.0 ** -6)


def test_six_piece_pairs_are_at_least_as_accurate_as_fp32_products():
    rng = np.random.default_rng(2)
    x, w, b = st.layer(32, 32, rng)
    ref = st.direct64(x, w[0])
    scale = float(np.abs(ref).max())
    err = {n: float(np.abs(st.wino(x, w[0], n).astype(np.float64) - ref).max()) / scale for n in (0, 3, 6, 9)}
    assert err[6] <= 1.2 * err[0], err               # six pairs: fp32-level (measured: better — the products are exact)
    assert err[9] <= err[6] * 1.25 and err[6] < 1e-5, err    # the last three pairs change nothing but the order of the fp32 sums
    assert err[3] > 10 * err[6], err                 # three pairs (hh, hm, mh) are NOT enough: ~1e-4
