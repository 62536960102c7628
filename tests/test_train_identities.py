"""CPU: the two algebraic identities round 4's training kernels rest on, in float64 against torch's own gradients.

  * Winograd-domain wgrad (csrc/train.hip wgrad_wino4_kernel): from Y = A^T [(G g G^T) . (B^T d B)] A per 4 x 4 output tile,
    dg = G^T [ sum over tiles of (B^T d B) . (A dY A^T) ] G — the matrices below are the ones the kernels hard-code (wg4_bt6,
    wg4_a4, wgrad_wino4_reduce_kernel) and the ones of the forward F(4x4,3x3) kernel (tests/wino4_ref.py).
  * polyphase dgrad of the stride-2 layers (read_amd/train.py _poly_fragments): every pixel parity of dx is the stride-1 dgrad of a
    3 x 3 pseudo-layer over the half-resolution dy, pseudo-weights taken from the taps [zero, W1, W3] / [W0, W2, zero] (4 x 4 layers)
    or [zero, W1, zero] / [W0, W2, zero] (3 x 3 layers) along each axis.
"""
import numpy as np
import torch
import torch.nn.functional as F

BT = np.array([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0],
               [0, 4, 0, -5, 0, 1]], np.float64)
G = np.array([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6],
              [0, 0, 1]], np.float64)
AT = np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], np.float64)


def test_winograd_domain_weight_gradient_identity():
    rng = np.random.default_rng(3)
    cin, cout, H, W = 3, 2, 8, 12                                   # whole 4 x 4 tiles, zero padding 1
    x = rng.standard_normal((cin, H, W))
    dy = rng.standard_normal((cout, H, W))
    xp = np.pad(x, ((0, 0), (1, 1), (1, 1)))
    dU = np.zeros((cout, cin, 6, 6))
    for ty in range(H // 4):
        for tx in range(W // 4):
            d = xp[:, 4 * ty:4 * ty + 6, 4 * tx:4 * tx + 6]           # the tile's 6 x 6 input patch (origin (4 ty - 1, 4 tx - 1))
            V = np.einsum("ar,crs,bs->cab", BT, d, BT)               # B^T d B per input channel
            M = np.einsum("pa,oab,qb->opq", AT.T, dy[:, 4 * ty:4 * ty + 4, 4 * tx:4 * tx + 4], AT.T)     # A dY A^T per output channel
            dU += M[:, None] * V[None]
    dg = np.einsum("xa,ocxn,nb->ocab", G, dU, G)                      # G^T dU G
    xt = torch.from_numpy(x)[None]
    ref = torch.nn.grad.conv2d_weight(xt, (cout, cin, 3, 3), torch.from_numpy(dy)[None], padding=1).numpy()
    assert np.abs(dg - ref).max() <= 1e-10 * np.abs(ref).max()
    # ... and the forward identity with the same three matrices (what makes them a valid triple)
    g = rng.standard_normal((cout, cin, 3, 3))
    y = F.conv2d(xt, torch.from_numpy(g), padding=1)[0].numpy()
    U = np.einsum("xa,ocab,nb->ocxn", G, g, G)
    for ty in range(H // 4):
        for tx in range(W // 4):
            V = np.einsum("ar,crs,bs->cab", BT, xp[:, 4 * ty:4 * ty + 6, 4 * tx:4 * tx + 6], BT)
            Y = np.einsum("pa,oab,qb->opq", AT, (U * V[None]).sum(1), AT)
            assert np.abs(Y - y[:, 4 * ty:4 * ty + 4, 4 * tx:4 * tx + 4]).max() <= 1e-10


def test_stride2_dgrad_is_four_stride1_dgrads_one_per_pixel_parity():
    torch.manual_seed(0)
    for k, taps in ((4, [[4, 1, 3], [0, 2, 4]]), (3, [[3, 1, 3], [0, 2, 3]])):       # the index tables of _poly_fragments (k = the zero tap)
        cin, cout, H, W = 6, 5, 8, 12
        x = torch.randn(1, cin, H, W, dtype=torch.double, requires_grad=True)
        w = torch.randn(cout, cin, k, k, dtype=torch.double)
        y = F.conv2d(x, w, stride=2, padding=1)
        assert tuple(y.shape[2:]) == (H // 2, W // 2)
        dy = torch.randn_like(y)
        y.backward(dy)
        t = torch.tensor(taps)
        A, B = t[[0, 0, 1, 1]][:, :, None], t[[0, 1, 0, 1]][:, None, :]
        wp = F.pad(w, (0, 1, 0, 1))[:, :, A, B].permute(2, 0, 1, 3, 4)           # (parity 2 py + px, cout, cin, 3, 3)
        dx = torch.zeros(cin, H, W, dtype=torch.double)
        for par in range(4):
            dx[:, par >> 1::2, par & 1::2] = F.conv_transpose2d(dy, wp[par], stride=1, padding=1)[0]     # stride-1 dgrad of a 3x3 layer
        assert float((dx - x.grad[0]).abs().max()) <= 1e-12
