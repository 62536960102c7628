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


def test_wave_autonomous_model_matches_conv2d():
    """The 16x16x4-MFMA kernel's lane maps (tests/wino16_ref.py): weights as the A operand with conv_f | conv_m stacked in the
    MFMA rows, tiles as columns, both 16-tile blocks, in-lane output transform."""
    from tests.wino16_ref import pack_w16, wino16_conv_model, wino16v2_conv_model
    rng = np.random.default_rng(0)
    cin, cout, H, W = 32, 40, 10, 20                      # two chunks, padded cout, partial tile blocks
    wf = rng.standard_normal((cout, cin, 3, 3)).astype(np.float32) * 0.2
    wm = rng.standard_normal((cout, cin, 3, 3)).astype(np.float32) * 0.2
    x = rng.standard_normal((H, W, cin)).astype(np.float32)
    f, m = wino16_conv_model(x, pack_w16(wf, wm), cin, cout)
    xt = torch.from_numpy(x).permute(2, 0, 1)[None]
    rf = F.conv2d(xt, torch.from_numpy(wf), padding=1)[0].permute(1, 2, 0).numpy()
    rm = F.conv2d(xt, torch.from_numpy(wm), padding=1)[0].permute(1, 2, 0).numpy()
    np.testing.assert_allclose(f, rf, rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(m, rm, rtol=1e-4, atol=1e-4)
    # version 2: the input transform shared through a swizzled LDS buffer (every slot written once, read where expected)
    f2, m2 = wino16v2_conv_model(x, pack_w16(wf, wm), cin, cout)
    np.testing.assert_allclose(f2, rf, rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(m2, rm, rtol=1e-4, atol=1e-4)


def test_library_packers_equal_the_models():
    """read_conv_pack_wino_host / read_conv_pack_w16_host (host code of libreadhip.so, no GPU needed) produce exactly the
    fragment orders of the NumPy models the kernels were checked against."""
    from read_amd import _lib
    from tests.wino16_ref import pack_w16
    L = _lib.lib()
    rng = np.random.default_rng(1)
    for cin, cout in ((16, 3), (32, 40), (48, 64)):
        wf = rng.standard_normal((cout, cin, 3, 3)).astype(np.float32)
        wm = rng.standard_normal((cout, cin, 3, 3)).astype(np.float32)
        n = L.read_conv_wino_floats(cin, cout)
        a, b = np.empty(n, np.float32), np.empty(n, np.float32)
        assert L.read_conv_pack_wino_host(cin, cout, wf.ctypes.data, wm.ctypes.data, a.ctypes.data) == 0
        assert L.read_conv_pack_w16_host(cin, cout, wf.ctypes.data, wm.ctypes.data, b.ctypes.data) == 0
        np.testing.assert_allclose(a, pack_wino(wf, wm), rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(b, pack_w16(wf, wm), rtol=1e-6, atol=1e-6)


def test_winograd_f4_model_and_packer():
    """F(4x4,3x3): the kernel's lane maps (tests/wino4_ref.py) against conv2d, and the library's host packer against the model's."""
    from read_amd import _lib
    from tests.wino4_ref import pack_w4, wino4_conv_model
    rng = np.random.default_rng(2)
    cin, cout, H, W = 32, 40, 11, 40                      # two chunks, padded cout, ragged 8 x 32 blocks
    wf = rng.standard_normal((cout, cin, 3, 3)).astype(np.float32) * 0.2
    wm = rng.standard_normal((cout, cin, 3, 3)).astype(np.float32) * 0.2
    x = rng.standard_normal((H, W, cin)).astype(np.float32)
    packed = pack_w4(wf, wm)
    f, m = wino4_conv_model(x, packed, cin, cout)
    xt = torch.from_numpy(x).permute(2, 0, 1)[None]
    rf = F.conv2d(xt, torch.from_numpy(wf), padding=1)[0].permute(1, 2, 0).numpy()
    rm = F.conv2d(xt, torch.from_numpy(wm), padding=1)[0].permute(1, 2, 0).numpy()
    np.testing.assert_allclose(f, rf, rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(m, rm, rtol=2e-4, atol=2e-4)
    L = _lib.lib()
    n = L.read_conv_w4_floats(cin, cout)
    assert n == packed.size
    got = np.empty(n, np.float32)
    assert L.read_conv_pack_w4_host(cin, cout, wf.ctypes.data, wm.ctypes.data, got.ctypes.data) == 0
    np.testing.assert_allclose(got, packed, rtol=1e-6, atol=1e-7)


def test_split_operand_f4_model_and_packer():
    """Split-operand F(4x4,3x3) on the f16 matrix cores: the kernel's lane maps and piece arithmetic (tests/wino4h_ref.py) against
    conv2d at the fp32 kernel's tolerance, the library's host packer against the model's BIT FOR BIT (halfs and row scales), and
    the properties the arithmetic rests on: max |U s| in [2^14, 2^15), hi + lo = U s to 2^-22, the swizzle bank-conflict free."""
    from read_amd import _lib
    from tests.wino4h_ref import filter_transform4_f64, pack_w4h, pack_w4h_blob, row_scale_exp, wino4h_conv_model
    rng = np.random.default_rng(6)
    cin, cout, H, W = 64, 40, 11, 40                      # two 32-channel chunks, padded cout, ragged 8 x 32 blocks
    wf = rng.standard_normal((cout, cin, 3, 3)).astype(np.float32) * 0.2
    wm = rng.standard_normal((cout, cin, 3, 3)).astype(np.float32) * 0.05
    wm[3] = 0.0                                           # an all-zero row keeps scale 1
    x = rng.standard_normal((H, W, cin)).astype(np.float32)
    halfs, inv = pack_w4h(wf, wm)
    f, m = wino4h_conv_model(x, halfs, inv, cin, cout)
    xt = torch.from_numpy(x).permute(2, 0, 1)[None]
    rf = F.conv2d(xt, torch.from_numpy(wf), padding=1)[0].permute(1, 2, 0).numpy()
    rm = F.conv2d(xt, torch.from_numpy(wm), padding=1)[0].permute(1, 2, 0).numpy()
    np.testing.assert_allclose(f, rf, rtol=1e-4, atol=1e-4)          # (the fp32 model: 2e-4; what is left is the fp32 round-off of the transforms)
    np.testing.assert_allclose(m, rm, rtol=1e-4, atol=1e-4)
    L = _lib.lib()
    n = L.read_conv_w4h_floats(cin, cout)
    blob = pack_w4h_blob(wf, wm)
    assert n == blob.size and L.read_conv_w4h_floats(48, 32) == 0     # whole 32-channel chunks only
    got = np.zeros(n, np.float32)
    assert L.read_conv_pack_w4h_host(cin, cout, wf.ctypes.data, wm.ctypes.data, got.ctypes.data) == 0
    assert np.array_equal(got.view(np.uint32), blob.view(np.uint32)), "library packer != model packer"
    U = filter_transform4_f64(wf)
    ex = row_scale_exp(U)
    top = np.abs(np.ldexp(U, ex[None, None, None, :])).max(axis=(0, 1, 2))
    assert np.all((top >= 2.0 ** 14) & (top < 2.0 ** 15))
    Us = np.ldexp(U, ex[None, None, None, :])
    hi = Us.astype(np.float16).astype(np.float64)
    lo = (Us - hi).astype(np.float16).astype(np.float64)
    assert np.abs(hi + lo - Us).max() <= 2.0 ** -22 * 2.0 ** 15
    assert inv[1, 3] == 1.0 and inv[0, 0] == np.float32(2.0 ** -float(ex[0]))
    # ds_read_b128 is served in four 16-lane groups, each must touch 16 different 16-byte slots of a 256-byte bank row
    lanes = np.arange(64)
    t, kl = lanes & 15, lanes >> 4
    addr16 = t * 4 + (kl ^ ((-(t >> 2)) & 3))             # 16-byte units inside one (frequency, piece) block
    for grp in ([0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]):
        for half in (0, 32):
            assert len({int(addr16[l + half]) % 16 for l in grp}) == 16


def test_direct_split_operand_packer_and_arithmetic():
    """The direct split-operand 3x3 kernel's operand (tests/d3h_ref.py): the library's host packer against the NumPy restatement BIT FOR
    BIT (halfs and row scales), the three-piece-pair arithmetic against conv2d at the DIRECT kernels' tolerance, and the scale rule."""
    from read_amd import _lib
    from tests.d3h_ref import pack_d3h_blob, row_scale_exp, split_conv_model
    rng = np.random.default_rng(8)
    cin, cout, H, W = 64, 40, 9, 21
    wf = rng.standard_normal((cout, cin, 3, 3)).astype(np.float32) * 0.1
    wm = rng.standard_normal((cout, cin, 3, 3)).astype(np.float32) * 0.03
    wm[5] = 0.0
    L = _lib.lib()
    blob = pack_d3h_blob(wf, wm)
    n = L.read_conv_d3h_floats(cin, cout)
    assert n == blob.size == cin * 18 * 64 + 2 * 64 and L.read_conv_d3h_floats(48, 32) == 0
    got = np.zeros(n, np.float32)
    assert L.read_conv_pack_d3h_host(cin, cout, wf.ctypes.data, wm.ctypes.data, got.ctypes.data) == 0
    assert np.array_equal(got.view(np.uint32), blob.view(np.uint32)), "library packer != model packer"
    ex = row_scale_exp(wf)
    top = np.abs(np.ldexp(wf.astype(np.float64), ex[:, None, None, None])).max(axis=(1, 2, 3))
    assert np.all((top >= 2.0 ** 14) & (top < 2.0 ** 15))
    x = rng.standard_normal((H, W, cin)).astype(np.float32)
    ref = F.conv2d(torch.from_numpy(x).permute(2, 0, 1)[None], torch.from_numpy(wf), padding=1)[0].permute(1, 2, 0).numpy()
    np.testing.assert_allclose(split_conv_model(x, wf), ref, rtol=2e-5, atol=2e-5)


def test_split_operand_1x1_packer_and_arithmetic():
    """The 1x1 layers' split operand (gated_conv_pxh_kernel; read_conv_pack_dkh_host with ksize 1): the library's host packer against
    the NumPy restatement BIT FOR BIT for whole and padded channel groups, the size rule, and the three-piece-pair arithmetic against
    conv2d at the fp32 kernels' tolerance — also for activations of 3000 and of 1e-3 (the low piece is scaled by 2^11)."""
    from read_amd import _lib
    from tests.d3h_ref import pack_d1h_blob, split_1x1_model
    rng = np.random.default_rng(9)
    L = _lib.lib()
    assert L.read_conv_dkh_floats(24, 32, 1) == 0 and L.read_conv_dkh_floats(8, 32, 1) == 0 and L.read_conv_dkh_floats(16, 4, 1) == 16 * 64 + 64
    for cin, cout in ((16, 32), (64, 56), (128, 248), (48, 4)):
        wf = rng.standard_normal((cout, cin, 1, 1)).astype(np.float32) * 0.1
        wm = rng.standard_normal((cout, cin, 1, 1)).astype(np.float32) * 0.03
        wm[1] = 0.0
        blob = pack_d1h_blob(wf, wm)
        cp = (cout + 31) // 32 * 32
        n = L.read_conv_dkh_floats(cin, cout, 1)
        assert n == blob.size == cin * 2 * cp + 2 * cp
        got = np.zeros(n, np.float32)
        assert L.read_conv_pack_dkh_host(cin, cout, 1, wf.ctypes.data, wm.ctypes.data, got.ctypes.data) == 0
        assert np.array_equal(got.view(np.uint32), blob.view(np.uint32)), "library packer != model packer"
        for scale in (1.0, 3000.0, 1e-3):
            x = rng.standard_normal((7, 13, cin)).astype(np.float32) * np.float32(scale)
            ref = F.conv2d(torch.from_numpy(x).permute(2, 0, 1)[None].double(), torch.from_numpy(wf).double())[0].permute(1, 2, 0).numpy()
            np.testing.assert_allclose(split_1x1_model(x, wf), ref, rtol=2e-5, atol=2e-5 * scale)


def test_split_operand_implicit_gemm_packer():
    """read_conv_pack_t3h_host: the 3x3 weights of a layer with 8, 16 or 32 input channels as the matrix W'[cout][tap * Cin + ci], zero-padded
    to whole k16 steps, in the 1x1 operand's order — against the NumPy restatement bit for bit; the size rule; the arithmetic of the
    implicit GEMM (zero padding at the border) against conv2d."""
    from read_amd import _lib
    from tests.d3h_ref import pack_d1h_blob, split_conv_model_taps
    rng = np.random.default_rng(10)
    L = _lib.lib()
    assert L.read_conv_t3h_floats(64, 32) == 0 and L.read_conv_t3h_floats(24, 32) == 0 and L.read_conv_t3h_floats(8, 32) == 80 * 64 + 64
    for cin, cout in ((8, 32), (8, 16), (16, 56), (32, 3)):
        wf = rng.standard_normal((cout, cin, 3, 3)).astype(np.float32) * 0.1
        wm = rng.standard_normal((cout, cin, 3, 3)).astype(np.float32) * 0.03
        K = (9 * cin + 15) // 16 * 16
        mat = lambda w: np.concatenate([w.transpose(0, 2, 3, 1).reshape(cout, 9 * cin), np.zeros((cout, K - 9 * cin), np.float32)], 1)   # noqa: E731
        blob = pack_d1h_blob(mat(wf), mat(wm))
        n = L.read_conv_t3h_floats(cin, cout)
        assert n == blob.size
        got = np.zeros(n, np.float32)
        assert L.read_conv_pack_t3h_host(cin, cout, wf.ctypes.data, wm.ctypes.data, got.ctypes.data) == 0
        assert np.array_equal(got.view(np.uint32), blob.view(np.uint32)), "library packer != model packer"
        H, W = 6, 11
        x = rng.standard_normal((H, W, cin)).astype(np.float32)
        xp = np.zeros((H + 2, W + 2, cin), np.float32)
        xp[1:-1, 1:-1] = x
        cols = np.concatenate([xp[ky:ky + H, kx:kx + W] for ky in range(3) for kx in range(3)] + [np.zeros((H, W, K - 9 * cin), np.float32)], 2)
        ref = F.conv2d(torch.from_numpy(x).permute(2, 0, 1)[None].double(), torch.from_numpy(wf).double(), padding=1)[0].permute(1, 2, 0).numpy()
        np.testing.assert_allclose(split_conv_model_taps(cols, mat(wf)), ref, rtol=2e-5, atol=2e-5)
