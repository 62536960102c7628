"""Layer-level parity of the fused gated-conv MFMA kernel with the torch-fp32 oracle
(oracle.unet_torch.basic_conv == READ/models/unet.py:44-53 on CPU).

Tolerance: the kernel accumulates in exact fp32 (v_mfma_f32_32x32x2_f32) but in a different
order than oneDNN, so results agree to fp32 round-off of a K-long dot product:
|diff| <= 2e-5 * max(1, |ref|) for K <= 4320 at unit-scale data."""
import os
import re

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import unet_torch
from read_amd import synthetic
from read_amd.gated_conv import PackedGatedConv, bilinear_up4, config_names, gated_conv

pytestmark = pytest.mark.gpu

ATOL = RTOL = 2e-5


def _state(cin, cout, k, seed=0):
    return synthetic.make_unet_state([("L", cin, cout, k)], seed)


def _pack(st, src_channels):
    b = "L.block."
    return PackedGatedConv(st[b + "conv_f.weight"], st[b + "conv_f.bias"], st[b + "conv_m.weight"], st[b + "conv_m.bias"],
                           st[b + "norm.weight"], st[b + "norm.bias"], st[b + "norm.running_mean"],
                           st[b + "norm.running_var"], src_channels=src_channels)


def _nhwc(x_chw):
    return x_chw.permute(1, 2, 0).contiguous().cuda()


def _close(got_hwc, ref_chw, what, scale=1.0):
    got = got_hwc.cpu().permute(2, 0, 1)
    assert got.shape == ref_chw.shape, (got.shape, ref_chw.shape)
    err = (got - ref_chw).abs()
    tol = scale * (ATOL + RTOL * ref_chw.abs())
    bad = err > tol
    assert not bool(bad.any()), f"{what}: {int(bad.sum())} of {bad.numel()} off, max err {float(err.max()):.3e}"


def _cfg_params(name):
    m = re.match(r"k(\d)s(\d)c(\d+)_(?:wave_)?p(\d)q(\d)(?:m(\d)n(\d))?", name)
    return tuple(int(g) if g is not None else 1 for g in m.groups())


def test_every_tile_configuration(hip):
    """Each compiled (ksize, stride, chunk, tiling) configuration against torch on a ragged image
    (partial tiles in x and y, zero padding on all borders)."""
    torch.manual_seed(0)
    for ci, name in enumerate(config_names()):
        k, s, kc, P, QG, WM, WN = _cfg_params(name)
        groups = WN * QG
        for g_mult in ((1, 2) if groups == 4 else (1,)):               # grid.y > 1 path
            cout = 32 * groups * g_mult - (8 if kc == 16 else 5)         # non-multiple of 32 -> padded channels
            cin = {8: 24, 16: 48, 32: 64}[kc]
            H, W = (22, 44) if s == 1 else (24, 80)
            st = _state(cin, cout, k, seed=ci)
            x = torch.randn(cin, H, W)
            ref = unet_torch.basic_conv(st, "L", x[None], k, stride=s, elu=True)[0]
            got = gated_conv(_pack(st, [cin]), [(_nhwc(x), 0)], stride=s, elu=True, config=ci)
            # Winograd F(2x2,3x3) sums transformed terms of mixed sign: ~4x the round-off of the direct form
            _close(got, ref, f"config {ci} {name} cout={cout}", scale=5.0 if "wino" in name else 1.0)


def test_winograd_kernel_on_unet_shapes(hip):
    """The F(2x2,3x3) variant on the four dominant C->C shapes (ragged sizes: partial 8x16 blocks, odd
    rows/columns), with residual, against the direct torch convolution."""
    torch.manual_seed(11)
    ci = [i for i, n in enumerate(config_names()) if "wino" in n]
    assert ci, "no Winograd configuration compiled"
    for j, (c, H, W) in enumerate([(32, 37, 75), (64, 24, 48), (128, 9, 17), (256, 8, 16), (32, 1, 1), (32, 352, 1216)]):
        st = _state(c, c, 3, seed=300 + j)
        x = torch.randn(c, H, W)
        res = torch.randn(c, H, W)
        ref = unet_torch.basic_conv(st, "L", x[None], 3, elu=j % 2 == 0)[0] + res
        for cfg in ci + [-3]:        # row-per-wave kernel(s); -3 = the wave-autonomous 16x16x4 kernel (tests/wino16_ref.py)
            got = gated_conv(_pack(st, [c]), [(_nhwc(x), 0)], elu=j % 2 == 0, residual=_nhwc(res), config=cfg)
            _close(got, ref, f"winograd config {cfg} {c}->{c} {H}x{W}", scale=5.0)


def test_winograd_f4_kernel(hip):
    """Winograd F(4x4,3x3) (config -5; the automatic choice for C >= 128): the UNet's 128- and 256-channel shapes, ragged sizes
    (partial 8 x 32 blocks, tiny images, three and five channel groups), residual, ELU on / off, the FAM multiply, several units per workgroup.  Stated
    tolerance: 10x the direct kernels' (|diff| <= 2e-4 (1 + |ref|)): the F(4x4) transforms multiply by up to 8 and sum mixed signs."""
    torch.manual_seed(21)
    for j, (c, H, W) in enumerate([(128, 9, 17), (256, 8, 32), (128, 23, 70), (32, 5, 3), (64, 40, 100), (128, 88, 304), (256, 44, 152), (96, 14, 37), (160, 3, 65)]):
        st = _state(c, c, 3, seed=500 + j)
        x = torch.randn(c, H, W)
        res = torch.randn(c, H, W)
        ref = unet_torch.basic_conv(st, "L", x[None], 3, elu=j % 2 == 0)[0] + res
        got = gated_conv(_pack(st, [c]), [(_nhwc(x), 0)], elu=j % 2 == 0, residual=_nhwc(res), config=-5)
        _close(got, ref, f"winograd F(4x4) {c}->{c} {H}x{W}", scale=10.0)
    for c in (128, 256):                                   # FAM: x1 + BC(x1 * x2), automatic choice (C >= 128 -> F(4x4))
        x1, x2 = torch.randn(c, 12, 40), torch.randn(c, 12, 40)
        st = _state(c, c, 3, seed=c + 7)
        ref = x1 + unet_torch.basic_conv(st, "L", (x1 * x2)[None], 3, elu=False)[0]
        got = gated_conv(_pack(st, [c]), [(_nhwc(x1), 0)], elu=False, mul=_nhwc(x2), residual=_nhwc(x1))
        _close(got, ref, f"FAM through F(4x4) C={c}", scale=10.0)


def test_direct_split_operand_kernel(hip):
    """The DIRECT 3x3 kernel with split fp32 operands on the f16 matrix cores (config -8; round 6): every tap executed, no Winograd
    transform — the oracle and shapes of the F(4x4) tests at the DIRECT kernels' tolerance (|diff| <= 2e-5 (1 + |ref|): nothing
    multiplies by 8 on the way), ragged sizes, border units, several units per workgroup, ELU on / off, residual, FAM's multiply,
    a wide dynamic range (activations of 1e-3 and of 2e4: no transform amplifies the input, so the f16 range covers them)."""
    from read_amd import _lib
    import ctypes
    from read_amd.gated_conv import conv_desc
    torch.manual_seed(31)
    fam = _lib.lib().read_conv_kernel_family
    for j, (c, H, W) in enumerate([(128, 9, 17), (256, 8, 32), (128, 23, 70), (32, 5, 3), (64, 40, 100), (128, 88, 304), (256, 44, 152),
                                   (96, 14, 37), (160, 3, 65), (32, 64, 96), (64, 1, 1), (32, 41, 130), (32, 176, 608)]):
        st = _state(c, c, 3, seed=700 + j)
        x = torch.randn(c, H, W)
        res = torch.randn(c, H, W)
        ref = unet_torch.basic_conv(st, "L", x[None], 3, elu=j % 2 == 0)[0] + res
        pk = _pack(st, [c])
        assert pk.wpacked_d3h is not None
        got = gated_conv(pk, [(_nhwc(x), 0)], elu=j % 2 == 0, residual=_nhwc(res), config=-8)
        _close(got, ref, f"direct split-operand 3x3 {c}->{c} {H}x{W}")
    st = _state(64, 64, 3, seed=9)
    x1 = _nhwc(torch.randn(64, 12, 40))
    # automatic choice: FAM's x1 * x2 launches run here (11 us faster than the fp32 kernel), plain launches on the Winograd split-operand
    # kernel (5 - 15 % faster than this one); read_tuning_set("conv_d3h", 32) / config -8 send the plain ones here too
    assert fam(ctypes.byref(conv_desc(_pack(st, [64]), [(x1, 0)]))) == 5 and fam(ctypes.byref(conv_desc(_pack(st, [64]), [(x1, 0)], mul=x1))) == 6
    assert fam(ctypes.byref(conv_desc(_pack(st, [64]), [(x1, 0)], config=-8))) == 6
    for c in (64, 128, 256):                               # FAM: x1 + BC(x1 * x2) through the automatic choice
        x1, x2 = torch.randn(c, 12, 40), torch.randn(c, 12, 40)
        st = _state(c, c, 3, seed=c + 7)
        ref = x1 + unet_torch.basic_conv(st, "L", (x1 * x2)[None], 3, elu=False)[0]
        got = gated_conv(_pack(st, [c]), [(_nhwc(x1), 0)], elu=False, mul=_nhwc(x2), residual=_nhwc(x1))
        _close(got, ref, f"FAM through the direct split-operand kernel C={c}")
    for amp in (1e-3, 300.0, 2.0e4):
        st = _state(64, 64, 3, seed=77)
        x = torch.randn(64, 24, 40) * amp
        ref = unet_torch.basic_conv(st, "L", x[None], 3, elu=True)[0]
        got = gated_conv(_pack(st, [64]), [(_nhwc(x), 0)], elu=True, config=-8)
        _close(got, ref, f"direct split-operand 3x3, activations x {amp}", scale=max(1.0, amp))
    # STRIDE 2 (the encoder's down-sampling layers; gated_conv_d3h_s2_kernel: the patch as four parity planes): even, odd and ragged sizes
    for j, (cin, cout, H, W) in enumerate([(32, 64, 24, 80), (64, 128, 22, 46), (128, 256, 44, 152), (32, 64, 13, 37), (64, 64, 2, 2),
                                          (32, 128, 176, 608)]):
        st = _state(cin, cout, 3, seed=900 + j)
        x = torch.randn(cin, H, W)
        ref = unet_torch.basic_conv(st, "L", x[None], 3, stride=2, elu=j % 2 == 0)[0]
        pk = _pack(st, [cin])
        assert fam(ctypes.byref(conv_desc(pk, [(_nhwc(x), 0)], stride=2))) == 6
        got = gated_conv(pk, [(_nhwc(x), 0)], stride=2, elu=j % 2 == 0)
        _close(got, ref, f"direct split-operand 3x3 / stride 2 {cin}->{cout} {H}x{W}")
    # ... and the decoder's 4 x 4 / stride-2 layers (sixteen taps over the same four planes)
    for j, (cin, cout, H, W) in enumerate([(128, 64, 44, 152), (256, 128, 22, 76), (64, 64, 10, 18), (32, 64, 7, 33), (64, 32, 88, 304), (32, 96, 9, 21)]):
        st = _state(cin, cout, 4, seed=950 + j)
        x = torch.randn(cin, H, W)
        ref = unet_torch.basic_conv(st, "L", x[None], 4, stride=2, elu=j % 2 == 0)[0]
        pk = _pack(st, [cin])
        assert fam(ctypes.byref(conv_desc(pk, [(_nhwc(x), 0)], stride=2))) == 6
        got = gated_conv(pk, [(_nhwc(x), 0)], stride=2, elu=j % 2 == 0)
        _close(got, ref, f"direct split-operand 4x4 / stride 2 {cin}->{cout} {H}x{W}")
    # different output width than input (Cout != Cin) and a second group count
    for (cin, cout) in ((32, 64), (128, 32), (64, 96)):
        st = _state(cin, cout, 3, seed=cin + cout)
        x = torch.randn(cin, 19, 45)
        ref = unet_torch.basic_conv(st, "L", x[None], 3, elu=True)[0]
        got = gated_conv(_pack(st, [cin]), [(_nhwc(x), 0)], elu=True, config=-8)
        _close(got, ref, f"direct split-operand 3x3 {cin}->{cout}")


def test_winograd_f4_split_operand_kernel(hip):
    """Winograd F(4x4,3x3) with split fp32 operands on the f16 matrix cores (config -7; round 6): the same shapes, the same oracle
    and the SAME tolerance as the fp32-matrix-core kernel above — two f16 pieces per operand, three piece pairs per product, fp32
    accumulation (profiles/r6_f16split_probe.txt: more accurate than the fp32 MFMA chain) — plus: agreement with that kernel to
    a quarter of the tolerance, top / bottom / left / right border units, several units per workgroup, ELU on / off, FAM."""
    from read_amd import _lib
    torch.manual_seed(21)
    fam = _lib.lib().read_conv_kernel_family
    for j, (c, H, W) in enumerate([(128, 9, 17), (256, 8, 32), (128, 23, 70), (32, 5, 3), (64, 40, 100), (128, 88, 304), (256, 44, 152),
                                   (96, 14, 37), (160, 3, 65), (32, 64, 96), (64, 1, 1), (32, 41, 130)]):
        st = _state(c, c, 3, seed=500 + j)
        x = torch.randn(c, H, W)
        res = torch.randn(c, H, W)
        ref = unet_torch.basic_conv(st, "L", x[None], 3, elu=j % 2 == 0)[0] + res
        pk = _pack(st, [c])
        assert pk.wpacked_w4h is not None
        got = gated_conv(pk, [(_nhwc(x), 0)], elu=j % 2 == 0, residual=_nhwc(res), config=-7)
        _close(got, ref, f"split-operand F(4x4) {c}->{c} {H}x{W}", scale=10.0)
        f32 = gated_conv(pk, [(_nhwc(x), 0)], elu=j % 2 == 0, residual=_nhwc(res), config=-5)
        _close(got, f32.cpu().permute(2, 0, 1), f"split-operand vs fp32 F(4x4) {c}->{c} {H}x{W}", scale=2.5)
    # FAM's x1 * x2 is not taken by this kernel (its launches stay on the fp32 matrix cores): the family query says so
    from read_amd.gated_conv import conv_desc
    st = _state(64, 64, 3, seed=9)
    x1 = _nhwc(torch.randn(64, 12, 40))
    d_plain, d_fam = conv_desc(_pack(st, [64]), [(x1, 0)]), conv_desc(_pack(st, [64]), [(x1, 0)], mul=x1)
    import ctypes
    assert fam(ctypes.byref(d_plain)) == 5 and fam(ctypes.byref(d_fam)) == 6       # automatic: FAM goes to the direct split-operand kernel
    d_plain.config = d_fam.config = -7
    assert fam(ctypes.byref(d_plain)) == 5 and fam(ctypes.byref(d_fam)) != 5       # the Winograd split-operand kernel does not take FAM's multiply
    # a wide dynamic range: activations of 1e-3 and of 300 (transformed inputs up to ~3e4, below the f16 limit of 65504)
    for amp in (1e-3, 300.0):
        st = _state(64, 64, 3, seed=77)
        x = torch.randn(64, 24, 40) * amp
        ref = unet_torch.basic_conv(st, "L", x[None], 3, elu=True)[0]
        got = gated_conv(_pack(st, [64]), [(_nhwc(x), 0)], elu=True, config=-7)
        _close(got, ref, f"split-operand F(4x4), activations x {amp}", scale=10.0 * max(1.0, amp))
    # ... and past the f16 limit of the transformed input (|B^T d B| >= 65504: activations of ~1e4) the kernel fails LOUDLY — Inf / NaN,
    # never a plausible wrong number — while the fp32-matrix-core kernel (read_tuning_set("conv_w4h", 0) / config -5) takes them
    st = _state(64, 64, 3, seed=78)
    x = torch.randn(64, 24, 40) * 2.0e4
    ref = unet_torch.basic_conv(st, "L", x[None], 3, elu=True)[0]
    got = gated_conv(_pack(st, [64]), [(_nhwc(x), 0)], elu=True, config=-7)
    assert not bool(torch.isfinite(got).all()), "activations beyond the f16 range must not produce a finite-looking frame"
    got32 = gated_conv(_pack(st, [64]), [(_nhwc(x), 0)], elu=True, config=-5)
    _close(got32, ref, "fp32 F(4x4), activations x 2e4", scale=10.0 * 2.0e4)


@pytest.mark.skipif(os.environ.get("READ_AMD_TEST_W4X2") != "1" or os.environ.get("READ_HIP_DEBUG") != "1",
                    reason="the two-waves-per-SIMD F(4x4) kernel was measured slower in round 5 (profiles/r5_w4x2_ab.json) and is "
                           "compiled into the debug library only; READ_HIP_DEBUG=1 READ_AMD_TEST_W4X2=1 runs its parity test")
def test_winograd_f4_two_waves_per_simd_variant(hip):
    """knob conv_w4x2: the frequency-split F(4x4,3x3) kernel (eight waves per workgroup; DESIGN.md 12.1 d, tests/test_wino4x2_model.py)
    against torch and against the one-wave-per-SIMD kernel on the shapes of test_winograd_f4_kernel."""
    from read_amd import _lib
    L = _lib.lib()
    torch.manual_seed(22)
    try:
        for j, (c, H, W) in enumerate([(32, 8, 32), (32, 5, 3), (64, 40, 100), (128, 9, 17), (256, 8, 32), (128, 23, 70), (128, 88, 304),
                                       (256, 44, 152), (96, 14, 37), (160, 3, 65)]):
            st = _state(c, c, 3, seed=600 + j)
            x = torch.randn(c, H, W)
            res = torch.randn(c, H, W)
            ref = unet_torch.basic_conv(st, "L", x[None], 3, elu=j % 2 == 0)[0] + res
            pk = _pack(st, [c])
            outs = {}
            for knob in (0, 1):
                _lib.check(L.read_tuning_set(b"conv_w4x2", knob))
                outs[knob] = gated_conv(pk, [(_nhwc(x), 0)], elu=j % 2 == 0, residual=_nhwc(res), config=-5)
                _close(outs[knob], ref, f"F(4x4) conv_w4x2={knob} {c}->{c} {H}x{W}", scale=10.0)
            assert float((outs[0] - outs[1]).abs().max()) <= 2e-4 * (1 + float(ref.abs().max()))
    finally:
        _lib.check(L.read_tuning_set(b"conv_w4x2", 0))


def test_winograd_kernel_odd_channel_counts_and_strides(hip):
    """Edge cases of the Winograd kernel's transposed epilogue: Cout that is not a multiple of 4 (scalar store
    fallback, partial last channel quad), padded output rows (out_cstride > Cout, with and without fill), a
    single 16-channel chunk, images smaller than one 8x16 tile block, residual on odd channel counts."""
    torch.manual_seed(12)
    ci = [i for i, n in enumerate(config_names()) if "wino" in n][0]
    cases = [(16, 30, 9, 21, None, None), (16, 3, 16, 16, 4, 1.0), (32, 41, 5, 3, 44, None), (48, 64, 8, 16, 64, None),
             (16, 32, 1, 40, None, None), (32, 6, 10, 18, 8, 0.25)]
    for j, (cin, cout, H, W, cs, fill) in enumerate(cases):
        st = _state(cin, cout, 3, seed=400 + j)
        x = torch.randn(cin, H, W)
        res = torch.randn(cout, H, W) if j % 2 == 0 else None
        ref = unet_torch.basic_conv(st, "L", x[None], 3, elu=j % 2 == 1)[0]
        if res is not None:
            ref = ref + res
        width = cs if cs is not None else cout
        for cfg in (ci, -3):                                           # both F(2x2) Winograd kernels
            out = torch.full((H, W, width), -7.0, device="cuda")       # sentinel: untouched channels must stay
            got = gated_conv(_pack(st, [cin]), [(_nhwc(x), 0)], elu=j % 2 == 1, config=cfg, out=out, out_channels=width,
                             residual=_nhwc(res) if res is not None else None, fill=fill)
            _close(got[:, :, :cout].contiguous(), ref, f"winograd edge case {j} (config {cfg}): {cin}->{cout} {H}x{W} cs={cs}", scale=5.0)
            if width > cout:
                pad = got[:, :, cout:].cpu()
                want = fill if fill is not None else -7.0
                assert bool((pad == want).all()), f"case {j}: padded channels hold {pad.unique().tolist()}, want {want}"


def test_unet_layer_shapes_auto_config(hip):
    """The (cin, cout, k, stride) shapes the UNet actually runs, automatic configuration choice,
    elu on/off, residual add."""
    torch.manual_seed(1)
    shapes = [(8, 32, 3, 1), (8, 16, 3, 1), (16, 32, 1, 1), (32, 32, 3, 1), (32, 56, 1, 1), (64, 64, 3, 1),
              (64, 120, 1, 1), (128, 128, 3, 1), (128, 248, 1, 1), (256, 256, 3, 1), (32, 64, 3, 2), (64, 128, 3, 2),
              (128, 256, 3, 2), (256, 128, 4, 2), (128, 64, 4, 2), (64, 32, 4, 2), (32, 3, 3, 1), (256, 128, 1, 1)]
    for i, (cin, cout, k, s) in enumerate(shapes):
        H, W = (16, 48) if s == 1 else (16, 64)
        st = _state(cin, cout, k, seed=100 + i)
        x = torch.randn(cin, H, W)
        elu = i % 2 == 0
        ref = unet_torch.basic_conv(st, "L", x[None], k, stride=s, elu=elu)[0]
        res = torch.randn_like(ref) if (s == 1 and cin == cout) else None
        got = gated_conv(_pack(st, [cin]), [(_nhwc(x), 0)], stride=s, elu=elu,
                         residual=_nhwc(res) if res is not None else None)
        # C >= 128 3x3/s1 layers take Winograd F(4x4,3x3) by default: its stated tolerance (test_winograd_f4_kernel)
        _close(got, ref + (res if res is not None else 0), f"shape {cin}->{cout} k{k} s{s}",
               scale=10.0 if (k == 3 and s == 1 and cin >= 128) else 1.0)


def test_concat_and_nearest_resample_sources(hip):
    """AFF-style input: torch.cat of four tensors at four resolutions (unet.py:239-254)."""
    torch.manual_seed(2)
    H, W = 16, 32
    chans, shifts = [32, 64, 128, 256], [1, 0, -1, -2]                   # AFF1's pattern at 1/2 scale
    xs = [torch.randn(c, (H << sh) if sh > 0 else (H >> -sh), (W << sh) if sh > 0 else (W >> -sh))
          for c, sh in zip(chans, shifts)]
    cat = torch.cat([F.interpolate(x[None], size=(H, W), mode="nearest") for x in xs], 1)
    # F.interpolate(size=) and scale_factor= agree for these power-of-two ratios (src = floor(dst*ratio))
    st = _state(480, 64, 1, seed=7)
    ref = unet_torch.basic_conv(st, "L", cat, 1, elu=True)[0]
    got = gated_conv(_pack(st, chans), [(_nhwc(x), sh) for x, sh in zip(xs, shifts)], elu=True)
    _close(got, ref, "AFF 1x1 over 4 resampled sources")
    # SCM tail: cat[x (8 ch), main(x) (56 ch)] -> 8-channel chunks
    a, b = torch.randn(8, H, W), torch.randn(56, H, W)
    st = _state(64, 64, 1, seed=8)
    ref = unet_torch.basic_conv(st, "L", torch.cat([a, b])[None], 1, elu=False)[0]
    got = gated_conv(_pack(st, [8, 56]), [(_nhwc(a), 0), (_nhwc(b), 0)], elu=False)
    _close(got, ref, "SCM 1x1 over cat[8,56]")


def test_coarse_sources_as_pre_activation_addends(hip):
    """The AFF plan of read_unet: a 1x1 conv commutes with nearest up-sampling, so the inputs that live at coarser levels
    are multiplied at their own level by `linear` launches and enter the layer as a pre-activation addend
    (read_conv_desc.pre); same reference as the single 480-channel launch (unet.py:239-254).  Through the pixel-lane
    kernel (default) and through the LDS-tiled kernels (conv_px = 0)."""
    from read_amd import _lib
    torch.manual_seed(21)
    H, W = 24, 64
    chans = [32, 64, 128]
    xs = [torch.randn(c, H >> i, W >> i) for i, c in enumerate(chans)]            # fine, 1/2, 1/4
    cat = torch.cat([F.interpolate(x[None], size=(H, W), mode="nearest") for x in xs], 1)
    try:
        for px, pxh in ((1, 16), (3, 0), (0, 0)):                                   # split-operand pixel-lane kernel (default), fp32 pixel-lane, LDS-tiled
            _lib.check(_lib.lib().read_tuning_set(b"conv_px", px))
            _lib.check(_lib.lib().read_tuning_set(b"conv_pxh", pxh))
            for cout in (32, 64, 128):                                              # one, two, four channel groups
                st = _state(sum(chans), cout, 1, seed=30 + cout)
                ref = unet_torch.basic_conv(st, "L", cat, 1, elu=True)[0]
                b = "L.block."

                def part(c0, c1, own):
                    sub = dict(st)
                    for br in ("conv_f", "conv_m"):
                        sub[b + br + ".weight"] = np.ascontiguousarray(st[b + br + ".weight"][:, c0:c1])
                        if not own:
                            sub[b + br + ".bias"] = np.zeros(cout, np.float32)
                    return _pack(sub, [c1 - c0])

                q2 = gated_conv(part(96, 224, False), [(_nhwc(xs[2]), 0)], linear=True)                       # 1/4: [f | m]
                q1 = gated_conv(part(32, 96, False), [(_nhwc(xs[1]), 0)], linear=True, pre=(q2, 0, cout, 1))  # 1/2, + up(q2)
                assert q1.shape == (H // 2, W // 2, 2 * cout)
                got = gated_conv(part(0, 32, True), [(_nhwc(xs[0]), 0)], elu=True, pre=(q1, 0, cout, 1))
                _close(got, ref, f"AFF split over three levels, Cout={cout}, conv_px={px}, conv_pxh={pxh}")
                lin = F.conv2d(xs[2][None], torch.as_tensor(np.ascontiguousarray(st[b + "conv_f.weight"][:, 96:224])))[0]
                _close(q2[:, :, :cout].contiguous(), lin, "linear 1x1 partial sum")
    finally:
        _lib.check(_lib.lib().read_tuning_set(b"conv_px", 1))
        _lib.check(_lib.lib().read_tuning_set(b"conv_pxh", 16))


def test_bilinear_upsampled_source_as_pre_activation_addend(hip):
    """Convs.k of read_unet (round 5): a 1x1 conv commutes with BILINEAR up-sampling too, so the half of
    cat[nn.Upsample(x4, bilinear)(fe), r] -> 1x1 (unet.py:261-262,269-270,277-278) that multiplies the up-sampled tensor is applied
    to fe at 1/16 of the pixels by a `linear` launch and enters the gated launch as a bilinear pre-activation addend
    (read_conv_desc.pre_bilinear): the up-sampled tensor is never written.  Reference: torch's own Upsample + the full conv.
    Ragged sizes (image edges clamp), all three channel counts of the network, and the refusals."""
    from read_amd import _lib
    torch.manual_seed(23)
    up = torch.nn.Upsample(scale_factor=4, mode="bilinear", align_corners=False)
    for (c, h, w) in ((128, 5, 19), (64, 11, 38), (32, 22, 76), (32, 1, 1), (64, 3, 2), (-128, 5, 19), (-32, 9, 7)):
        # c < 0: the same through the fp32 pixel-lane kernel (conv_pxh = 0); default: the split-operand pixel-lane kernel
        _lib.check(_lib.lib().read_tuning_set(b"conv_pxh", 0 if c < 0 else 16))
        c = abs(c)
        fe, r = torch.randn(c, h, w), torch.randn(c, 4 * h, 4 * w)
        st = _state(2 * c, c, 1, seed=40 + c + h)
        ref = unet_torch.basic_conv(st, "L", torch.cat([up(fe[None]), r[None]], 1), 1, elu=True)[0]
        b = "L.block."

        def part(c0, c1, own):
            sub = dict(st)
            for br in ("conv_f", "conv_m"):
                sub[b + br + ".weight"] = np.ascontiguousarray(st[b + br + ".weight"][:, c0:c1])
                if not own:
                    sub[b + br + ".bias"] = np.zeros(c, np.float32)
            return _pack(sub, [c1 - c0])

        q = gated_conv(part(0, c, False), [(_nhwc(fe), 0)], linear=True)                      # (h, w, 2c): [f | m] at fe's level
        got = gated_conv(part(c, 2 * c, True), [(_nhwc(r), 0)], elu=True, pre=(q, 0, c, 2, True))
        _close(got, ref, f"bilinear addend C={c} {h}x{w}")
        res = torch.randn(c, 4 * h, 4 * w)                                                     # + residual through the same epilogue
        got = gated_conv(part(c, 2 * c, True), [(_nhwc(r), 0)], elu=False, pre=(q, 0, c, 2, True), residual=_nhwc(res))
        ref2 = unet_torch.basic_conv(st, "L", torch.cat([up(fe[None]), r[None]], 1), 1, elu=False)[0] + res
        _close(got, ref2, f"bilinear addend + residual C={c} {h}x{w}")
    _lib.check(_lib.lib().read_tuning_set(b"conv_pxh", 16))
    # refusals: another shift, a 3x3 layer
    fe, r = torch.randn(32, 4, 4), torch.randn(32, 16, 16)
    st = _state(32, 32, 1, seed=3)
    q = gated_conv(_pack(st, [32]), [(_nhwc(fe), 0)], linear=True)
    with pytest.raises(_lib.ReadHipError, match="pre_bilinear"):
        gated_conv(_pack(st, [32]), [(_nhwc(r[:, :8, :8].contiguous()), 0)], pre=(q, 0, 32, 1, True))
    st3 = _state(32, 32, 3, seed=4)
    with pytest.raises(_lib.ReadHipError):
        gated_conv(_pack(st3, [32]), [(_nhwc(r), 0)], pre=(q, 0, 32, 2, True))


def test_pixel_lane_kernel_for_1x1_layers(hip):
    """config=-2 forces the pixel-lane 1x1 kernel (weights as the MFMA A operand, LDS-resident; activations straight
    from NHWC memory): the 1x1 shapes of the network, ragged pixel counts, resampled / concatenated sources, residual,
    padded channel groups (Cout = 56) and both accumulator widths."""
    from read_amd import _lib
    torch.manual_seed(31)
    cases = [  # (source channels, shifts, Cout, elu, residual)
        ([16], [0], 32, True, False), ([32], [0], 56, True, False), ([8, 56], [0, 0], 64, False, False),
        ([128], [0], 248, True, False), ([8, 248], [0, 0], 256, False, True), ([128, 128], [0, 0], 128, True, True),
        ([32, 64, 128], [2, 1, 0], 128, True, False), ([32, 64], [1, 0], 64, True, False), ([64, 64], [0, 0], 32, True, False),
    ]
    try:
        for width in (3, 4):
            _lib.check(_lib.lib().read_tuning_set(b"conv_px", width))
            for chans, shifts, cout, elu, with_res in cases:
                H, W = 12, 44                                           # 528 pixels: ragged last tile, rows not a multiple of 32
                xs = [torch.randn(c, H << sh, W << sh) for c, sh in zip(chans, shifts)]
                cat = torch.cat([F.interpolate(x[None], size=(H, W), mode="nearest") for x in xs], 1)
                st = _state(sum(chans), cout, 1, seed=sum(chans) + cout)
                ref = unet_torch.basic_conv(st, "L", cat, 1, elu=elu)[0]
                res = torch.randn(cout, H, W) if with_res else None
                got = gated_conv(_pack(st, chans), [(_nhwc(x), sh) for x, sh in zip(xs, shifts)], elu=elu, config=-2,
                                 residual=_nhwc(res) if with_res else None)
                _close(got, ref + (res if with_res else 0), f"px width {width}: {chans} -> {cout}")
    finally:
        _lib.check(_lib.lib().read_tuning_set(b"conv_px", 1))
    with pytest.raises(_lib.ReadHipError):                               # 3x3 layers do not qualify
        st = _state(32, 32, 3, seed=1)
        gated_conv(_pack(st, [32]), [(_nhwc(torch.randn(32, 8, 32)), 0)], config=-2)


def test_split_operand_pixel_lane_kernel_for_1x1_layers(hip):
    """gated_conv_pxh_kernel (round 6): the 1x1 layers with split fp32 operands on the f16 matrix cores (config=-10 forces it, the
    default takes it): every 1x1 shape of the network — SCM tails incl. cat[x(8), main(P - 8)] whose half-waves read different
    sources, padded channel groups (Cout = 56 / 120 / 248), Convs.k, the AFF pieces —, ragged pixel counts, resampled sources, residual,
    linear launches, all three tile shapes (two groups per wave; two pixel tiles per wave at frame size; one and one on small images).
    Same reference and the same tolerance as the fp32 kernels."""
    import ctypes
    from read_amd import _lib
    from read_amd.gated_conv import conv_desc
    fam = _lib.lib().read_conv_kernel_family
    torch.manual_seed(33)
    cases = [  # (source channels, shifts, Cout, elu, residual, (H, W))
        ([16], [0], 32, True, False, (12, 44)), ([32], [0], 56, True, False, (12, 44)), ([8, 56], [0, 0], 64, False, False, (12, 44)),
        ([64], [0], 120, True, False, (12, 44)), ([8, 120], [0, 0], 128, False, False, (11, 38)),
        ([128], [0], 248, True, False, (12, 44)), ([8, 248], [0, 0], 256, False, True, (12, 44)), ([128, 128], [0, 0], 128, True, True, (12, 44)),
        ([32, 64, 128], [2, 1, 0], 128, True, False, (12, 44)), ([32, 64], [1, 0], 64, True, False, (12, 44)),
        ([64, 64], [0, 0], 32, True, False, (12, 44)), ([24, 8, 16], [0, 0, 0], 32, True, False, (9, 33)),
        ([32, 32], [0, 0], 32, True, True, (304, 448)),                 # 136 K pixels: two pixel tiles per wave
        ([256], [0], 256, False, False, (44, 152)), ([96], [0], 64, True, False, (40, 100)),
    ]
    for chans, shifts, cout, elu, with_res, (H, W) in cases:
        xs = [torch.randn(c, H << sh, W << sh) for c, sh in zip(chans, shifts)]
        cat = torch.cat([F.interpolate(x[None], size=(H, W), mode="nearest") for x in xs], 1)
        st = _state(sum(chans), cout, 1, seed=sum(chans) + cout)
        ref = unet_torch.basic_conv(st, "L", cat, 1, elu=elu)[0]
        res = torch.randn(cout, H, W) if with_res else None
        pk = _pack(st, chans)
        assert pk.wpacked_d3h is not None
        srcs = [(_nhwc(x), sh) for x, sh in zip(xs, shifts)]
        assert fam(ctypes.byref(conv_desc(pk, srcs))) == 7 and fam(ctypes.byref(conv_desc(pk, srcs, config=-10))) == 7
        for cfg in (-10, -1):
            got = gated_conv(pk, srcs, elu=elu, config=cfg, residual=_nhwc(res) if with_res else None)
            _close(got, ref + (res if with_res else 0), f"pxh {chans} -> {cout} at {H}x{W}, config {cfg}")
        lin = gated_conv(pk, srcs, linear=True, config=-10)            # [conv_f + b_f | conv_m + b_m]
        b = "L.block."
        ref_f = F.conv2d(cat, torch.as_tensor(st[b + "conv_f.weight"]), torch.as_tensor(st[b + "conv_f.bias"]))[0]
        ref_m = F.conv2d(cat, torch.as_tensor(st[b + "conv_m.weight"]), torch.as_tensor(st[b + "conv_m.bias"]))[0]
        _close(lin[:, :, :cout].contiguous(), ref_f, f"pxh linear f {chans} -> {cout}")
        _close(lin[:, :, cout:].contiguous(), ref_m, f"pxh linear m {chans} -> {cout}")
    # large activations (the f16 pieces cover |x| < 65504) and tiny ones (the low piece is scaled by 2^11: no f16 underflow)
    for scale in (3000.0, 1e-3):
        x = torch.randn(64, 12, 44) * scale
        st = _state(64, 64, 1, seed=5)
        ref = unet_torch.basic_conv(st, "L", x[None], 1, elu=True)[0]
        got = gated_conv(_pack(st, [64]), [(_nhwc(x), 0)], elu=True, config=-10)
        _close(got, ref, f"pxh at input scale {scale}", scale=max(1.0, scale / 30.0))
    # the knob: conv_pxh = 0 sends the layer back to the fp32 kernels; shapes the kernel does not take
    try:
        _lib.check(_lib.lib().read_tuning_set(b"conv_pxh", 0))
        st = _state(64, 64, 1, seed=6)
        x = _nhwc(torch.randn(64, 8, 32))
        assert fam(ctypes.byref(conv_desc(_pack(st, [64]), [(x, 0)]))) == 0
    finally:
        _lib.check(_lib.lib().read_tuning_set(b"conv_pxh", 16))
    st = _state(480, 64, 1, seed=7)                                      # more than 256 input channels: no operand, fp32 kernels
    assert _pack(st, [480]).wpacked_d3h is None or fam(ctypes.byref(conv_desc(_pack(st, [480]), [(_nhwc(torch.randn(480, 8, 32)), 0)]))) == 0
    with pytest.raises(_lib.ReadHipError):
        gated_conv(_pack(st, [480]), [(_nhwc(torch.randn(480, 8, 32)), 0)], config=-10)
    with pytest.raises(_lib.ReadHipError):                               # 3x3 layers do not qualify
        st3 = _state(32, 32, 3, seed=1)
        gated_conv(_pack(st3, [32]), [(_nhwc(torch.randn(32, 8, 32)), 0)], config=-10)


def test_split_operand_implicit_gemm_for_3x3_layers_over_few_channels(hip):
    """The TAPS form of gated_conv_pxh_kernel (round 6): 3x3 / stride-1 layers over one source of 8, 16 or 32 channels as an implicit GEMM with
    split operands (k = tap * Cin + channel).  By default the layers over the 8-channel descriptor pyramid (feat_extract.0, SCM.main.0:
    family 8); config = -11 / read_tuning_set("conv_t3h", 32) for 16 and 32 channels.  Zero padding at the border (ragged sizes, one-row
    and one-column images), residual, linear launches, all tile shapes; same reference and tolerance as the fp32 kernels."""
    import ctypes
    from read_amd import _lib
    from read_amd.gated_conv import conv_desc
    fam = _lib.lib().read_conv_kernel_family
    torch.manual_seed(35)
    cases = [  # (Cin, Cout, elu, residual, (H, W))
        (8, 32, True, False, (37, 61)), (8, 16, True, False, (22, 76)), (8, 64, True, False, (11, 38)), (8, 32, False, True, (9, 33)),
        (8, 32, True, False, (1, 40)), (8, 32, True, False, (40, 1)), (8, 32, True, False, (3, 3)),
        (8, 32, True, False, (304, 448)),                               # 136 K pixels: two pixel tiles per wave
        (16, 32, True, False, (21, 45)), (32, 32, True, True, (21, 45)), (32, 64, False, False, (13, 70)), (32, 4, True, False, (21, 45)),
    ]
    for cin, cout, elu, with_res, (H, W) in cases:
        x = torch.randn(cin, H, W)
        st = _state(cin, cout, 3, seed=cin + cout + H)
        ref = unet_torch.basic_conv(st, "L", x[None], 3, elu=elu)[0]
        res = torch.randn(cout, H, W) if with_res else None
        pk = _pack(st, [cin])
        assert pk.wpacked_t3h is not None
        srcs = [(_nhwc(x), 0)]
        assert fam(ctypes.byref(conv_desc(pk, srcs, config=-11))) == 8
        assert (fam(ctypes.byref(conv_desc(pk, srcs))) == 8) == (cin == 8 and H * W >= 16384)      # the default takes it from 16 K pixels on
        for cfg in ((-11, -1) if cin == 8 else (-11,)):
            got = gated_conv(pk, srcs, elu=elu, config=cfg, residual=_nhwc(res) if with_res else None)
            _close(got, ref + (res if with_res else 0), f"t3h {cin} -> {cout} at {H}x{W}, config {cfg}")
        lin = gated_conv(pk, srcs, linear=True, config=-11)
        b = "L.block."
        ref_f = F.conv2d(x[None], torch.as_tensor(st[b + "conv_f.weight"]), torch.as_tensor(st[b + "conv_f.bias"]), padding=1)[0]
        _close(lin[:, :, :cout].contiguous(), ref_f, f"t3h linear f {cin} -> {cout}")
    try:                                                                 # the knob
        st = _state(8, 32, 3, seed=2)
        x = _nhwc(torch.randn(8, 128, 160))
        assert fam(ctypes.byref(conv_desc(_pack(st, [8]), [(x, 0)]))) == 8
        _lib.check(_lib.lib().read_tuning_set(b"conv_t3h", 0))
        assert fam(ctypes.byref(conv_desc(_pack(st, [8]), [(x, 0)]))) == 0
        _lib.check(_lib.lib().read_tuning_set(b"conv_t3h", 32))
        st32 = _state(32, 64, 3, seed=3)
        assert fam(ctypes.byref(conv_desc(_pack(st32, [32]), [(_nhwc(torch.randn(32, 128, 160)), 0)]))) == 8
    finally:
        _lib.check(_lib.lib().read_tuning_set(b"conv_t3h", 8))
    with pytest.raises(_lib.ReadHipError):                               # 64 input channels: no operand
        st64 = _state(64, 64, 3, seed=4)
        gated_conv(_pack(st64, [64]), [(_nhwc(torch.randn(64, 8, 32)), 0)], config=-11)


def test_fam_multiply_and_residual(hip):
    """FAM: x1 + BC(x1*x2) (unet.py:114-117) in one launch."""
    torch.manual_seed(3)
    for c in (64, 128, 256):
        x1, x2 = torch.randn(c, 12, 40), torch.randn(c, 12, 40)
        st = _state(c, c, 3, seed=c)
        ref = x1 + unet_torch.basic_conv(st, "L", (x1 * x2)[None], 3, elu=False)[0]
        got = gated_conv(_pack(st, [c]), [(_nhwc(x1), 0)], elu=False, mul=_nhwc(x2), residual=_nhwc(x1))
        _close(got, ref, f"FAM C={c}", scale=10.0 if c >= 128 else 5.0)


def test_rgba_output_fill(hip):
    st = _state(32, 3, 3, seed=9)
    x = torch.randn(32, 16, 32)
    ref = unet_torch.basic_conv(st, "L", x[None], 3, elu=False)[0]
    got = gated_conv(_pack(st, [32]), [(_nhwc(x), 0)], elu=False, out_channels=4, fill=1.0)
    _close(got[:, :, :3].contiguous(), ref, "rgb")
    assert bool((got[:, :, 3] == 1.0).all())


def test_small_cout_vector_pipe_kernel(hip):
    """The 32 -> 3 output layer (READ/models/unet.py:205 feat_extract[5]) on the vector pipe (config=-6 forces it; knob conv_sc
    A/B): ragged image (partial 8x32 tiles in x and y), Cout 1..4, elu on and off, the RGBA fill, against torch and against the
    MFMA kernel it replaces (same tolerance: a 288-long fp32 dot product in another order)."""
    from read_amd import _lib
    L = _lib.lib()
    torch.manual_seed(17)
    for (H, W) in ((16, 32), (21, 45), (8, 96)):
        for cout in (3, 4, 1):
            for elu in (False, True):
                st = _state(32, cout, 3, seed=cout + H)
                x = torch.randn(32, H, W)
                ref = unet_torch.basic_conv(st, "L", x[None], 3, elu=elu)[0]
                pk = _pack(st, [32])
                got = gated_conv(pk, [(_nhwc(x), 0)], elu=elu, config=-6)
                _close(got, ref, f"small-Cout {H}x{W} cout={cout} elu={elu}")
    st = _state(32, 3, 3, seed=9)
    x = torch.randn(32, 24, 64)
    ref = unet_torch.basic_conv(st, "L", x[None], 3, elu=False)[0]
    pk = _pack(st, [32])
    try:
        frames = {}
        for on in (8, 16, 32, 0):                                        # channels per LDS phase; 0 = the MFMA kernel
            _lib.check(L.read_tuning_set(b"conv_sc", on))
            d_family = gated_conv(pk, [(_nhwc(x), 0)], elu=False, out_channels=4, fill=1.0)
            frames[on] = d_family
            _close(d_family[:, :, :3].contiguous(), ref, f"rgb conv_sc={on}")
            assert bool((d_family[:, :, 3] == 1.0).all())
        for other in (16, 32, 0):                                        # another order of the same 288 products per channel
            assert float((frames[8] - frames[other]).abs().max()) <= 2e-5 * 4
    finally:
        _lib.check(L.read_tuning_set(b"conv_sc", 8))
    with pytest.raises(_lib.ReadHipError):                               # wide layers do not qualify
        st = _state(32, 32, 3, seed=1)
        gated_conv(_pack(st, [32]), [(_nhwc(torch.randn(32, 8, 32)), 0)], config=-6)


def test_bilinear_up4(hip):
    x = torch.randn(64, 6, 10)
    ref = F.interpolate(x[None], scale_factor=4, mode="bilinear", align_corners=False)[0]
    got = bilinear_up4(_nhwc(x))
    torch.testing.assert_close(got.cpu().permute(2, 0, 1), ref, rtol=1e-6, atol=1e-6)


def test_wave_kernel_full_and_tail_units_at_frame_size(hip):
    """At 1216x352 the persistent wave kernel schedules whole rounds of 2-row units plus a tail of 1-row
    units; small test images only exercise the tail path.  Same k order as the workgroup-tiled kernel
    (already checked against the oracle), so the two must agree to the last bit or nearly so."""
    torch.manual_seed(5)
    names = config_names()
    for (c, H, W) in ((32, 352, 1216), (64, 176, 608)):
        st = _state(c, c, 3, seed=c)
        x = torch.randn(H, W, c, device="cuda")
        res = torch.randn(H, W, c, device="cuda")
        pk = _pack(st, [c])
        ref = gated_conv(pk, [(x, 0)], elu=True, residual=res, config=names.index("k3s1c16_p2q1m4n1f1b2"))
        for name in names:
            if name.startswith("k3s1c16_wave"):
                got = gated_conv(pk, [(x, 0)], elu=True, residual=res, config=names.index(name))
                err = float((got - ref).abs().max())
                assert err <= 1e-5, f"{name} at {W}x{H}x{c}: max diff {err:.3e}"


@pytest.mark.gpu
def test_mfma_rate_probe_reports_a_plausible_matrix_rate(hip):
    """read_mfma_f32_rate_probe (bench.py: roofline.mfma_sustained): the launch's flop count is what its grid executes, and the rate it
    measures lies between half of and the whole fp32 matrix peak of the guide (157 TF at 2.4 GHz; 146 measured)."""
    import ctypes as C
    from read_amd import _lib
    L = _lib.lib()
    scratch = torch.zeros(256 * 1024, dtype=torch.float32, device="cuda")
    fl = C.c_double(0.0)
    best = 0.0
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.check(L.read_mfma_f32_rate_probe(20000, scratch.data_ptr(), C.byref(fl), _lib.stream_ptr()), "read_mfma_f32_rate_probe")
        e1.record()
        e1.synchronize()
        best = max(best, fl.value / (e0.elapsed_time(e1) * 1e-3) / 1e12)
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    assert fl.value == cus * 4 * 20000 * 16 * 2048
    assert 80.0 < best < 160.0, best
    assert float(scratch.abs().max()) == 0.0                   # the probe writes nothing
