"""CPU: the Winograd F(2x2,3x3) index maps / transforms used by the HIP kernel, against torch conv2d."""
import numpy as np
import torch
import torch.nn.functional as F

from tests.wino_ref import filter_transform, pack_wino, wino_conv_model


def test_winograd_model_matches_conv2d():
    rng = np.random.default_rng(0)
    cin, cout, H, W = 16, 40, 10, 20                      # partial tile blocks in both directions, padded cout
    wf = rng.standard_normal((cout, cin, 3, 3)).astype(np.float32) * 0.2
    wm = rng.standard_normal((cout, cin, 3, 3)).astype(np.float32) * 0.2
    x = rng.standard_normal((H, W, cin)).astype(np.float32)
    f, m = wino_conv_model(x, pack_wino(wf, wm), cin, cout)
    xt = torch.from_numpy(x).permute(2, 0, 1)[None]
    rf = F.conv2d(xt, torch.from_numpy(wf), padding=1)[0].permute(1, 2, 0).numpy()
    rm = F.conv2d(xt, torch.from_numpy(wm), padding=1)[0].permute(1, 2, 0).numpy()
    np.testing.assert_allclose(f, rf, rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(m, rm, rtol=1e-4, atol=1e-4)


def test_filter_transform_shape_and_dc():
    w = np.ones((1, 1, 3, 3), np.float32)
    U = filter_transform(w)
    assert U.shape == (4, 4, 1, 1)
    assert abs(float(U[0, 0, 0, 0]) - 1.0) < 1e-6 and abs(float(U[1, 1, 0, 0]) - 2.25) < 1e-6
