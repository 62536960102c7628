"""CPU: the C-ABI library loads without a GPU, exports every symbol include/read_hip.h declares,
rejects bad arguments with READ_EINVAL + a message, and its host-side packers produce the layout the
kernels index.  No compute call is made."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from read_amd import _lib
from read_amd.gated_conv import kc_for
from tests.unet_spec import UNET_SPEC

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "read_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(read_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    L = C.CDLL(_lib.LIB_PATH)
    names = _declared_symbols()
    assert len(names) >= 30
    for n in names:
        assert hasattr(L, n), f"{n} declared in read_hip.h but not exported"
    assert set(names) == set(_lib.SIGNATURES), set(names) ^ set(_lib.SIGNATURES)
    assert _lib.lib().read_abi_version() == 3


def test_layer_table_matches_independent_spec():
    from read_amd.unet import layer_table, weight_spec
    assert weight_spec() == UNET_SPEC
    strides = {p: s for (p, _, _, _, s, _) in layer_table()}
    assert [p for p, s in strides.items() if s == 2] == ["feat_extract.1", "feat_extract.2", "feat_extract.3",
                                                         "feat_extract.4", "feat_extract.6", "feat_extract.7"]
    no_elu = {p for (p, _, _, _, _, e) in layer_table() if not e}
    assert "feat_extract.5" in no_elu and "Encoder.0.layers.0.main.1" in no_elu and "SCM0.conv" in no_elu
    assert "Encoder.0.layers.0.main.0" not in no_elu


def test_bad_arguments_fail_loudly():
    L = _lib.lib()
    # header + one key image + hi-z bounds + two seed images + the two depth-bound images of the cell path (frames alternate)
    # + the bins of its pass A (38 x 11 tiles of 32x32 pixels x 32 sub-bins: a counter and 256 records of 16 bytes each)
    px = 1216 * 352
    bins = 38 * 11 * 32
    want = 8192 + 8 * px * 8 + 304 * 88 * 4 + 4 * px * 4 + (bins * 4 + 255) // 256 * 256 + bins * 256 * 16
    assert L.read_splat_workspace_bytes(1, 1216, 352) == want
    assert L.read_splat_workspace_bytes(9, 1216, 352) == want
    assert L.read_splat_workspace_bytes(0, 10, 10) == 0
    rc = L.read_splat_forward(None, 10, None, 1, 64, 64, 5, None, None, None, 0, None)
    assert rc == -22 and b"xyz" in L.read_last_error()
    rc = L.read_gather_forward(None, 0, 8, 1, None, None, None, 0, None)
    assert rc == -22
    assert L.read_unet_workspace_bytes(100, 100) == 0                  # not a multiple of 16
    assert L.read_splat_hint_next_camera(None, None) == -22 and b"workspace" in L.read_last_error()
    assert L.read_splat_profile_last(None) == -22
    buf = (C.c_float * 5)()
    assert L.read_splat_profile_last(buf) == -22 and b"splat_prof" in L.read_last_error()      # no profiled frame yet
    assert L.read_mfma_f32_rate_probe(0, None, None, None) == -22 and b"read_mfma_f32_rate_probe" in L.read_last_error()
    with pytest.raises(_lib.ReadHipError):
        _lib.check(L.read_bilinear_up4(None, 4, 4, 8, None, None), "up4")
    d = _lib.ConvDesc()
    assert L.read_gated_conv_forward(C.byref(d), None) == -22
    # the release library has no knob that produces invalid results
    assert L.read_tuning_set(b"conv_ablate", 1) == -22
    for bad_mode in (2, 4, 5, 6):
        assert L.read_tuning_set(b"splat_mode", bad_mode) == -22
    st = _lib.tuning_state()
    assert st["splat_mode"] == 7 and st["splat_cells"] == 1 and st["splat_strips"] == 1 and st["splat_items"] == 4 and st["splat_lds"] == 1 and st["splat_kslot"] == 0 and st["conv_wino"] == 1 << 30 and "conv_ablate" not in st


def test_weight_packing_layout():
    """wpacked[(((chunk*taps+tap)*KK+kk)*NT+nt)*256 + lane*4 + j] = W{f|m}[cout][cin][tap] with
    cout = (nt//2)*32 + lane%32, cin = chunk*kc + kk*8 + 4*(lane//32) + j  (conv.hip header)."""
    L = _lib.lib()
    rng = np.random.default_rng(0)
    for (cin, cout, k, kc) in [(32, 32, 3, 16), (24, 40, 3, 8), (64, 56, 1, 8), (48, 3, 4, 16)]:
        wf = rng.standard_normal((cout, cin, k, k)).astype(np.float32)
        wm = rng.standard_normal((cout, cin, k, k)).astype(np.float32)
        n = L.read_conv_packed_floats(cin, cout, k)
        cp = (cout + 31) // 32 * 32
        assert n == cin * k * k * 2 * cp
        out = np.full(n, np.nan, np.float32)
        _lib.check(L.read_conv_pack_weights_host(cin, cout, k, kc, wf.ctypes.data, wm.ctypes.data, out.ctypes.data))
        NT, KK, taps = cp // 16, kc // 8, k * k
        o = out.reshape(cin // kc, taps, KK, NT, 64, 4)
        for _ in range(200):
            ch, tap, kk, nt, lane, j = (rng.integers(0, s) for s in o.shape)
            co, ci = (nt // 2) * 32 + lane % 32, ch * kc + kk * 8 + 4 * (lane // 32) + j
            w = wm if nt % 2 else wf
            want = w[co, ci, tap // k, tap % k] if co < cout else 0.0
            assert o[ch, tap, kk, nt, lane, j] == want
        assert L.read_conv_pack_weights_host(cin, cout, k, 16 if cin % 16 else 8 if cin % 8 else 16, None, None, None) == -22


def test_small_cout_weight_order():
    """read_conv_pack_sc_host: out[(tap * Cin + cin) * 8 + j] = Wf[j][cin][tap] (j < 4) | Wm[j - 4][cin][tap], channels >= Cout zero
    (the vector-pipe kernel of the 32 -> 3 output layer reads it with scalar loads); only Cin = 32, Cout <= 4 have the order."""
    L = _lib.lib()
    rng = np.random.default_rng(4)
    assert L.read_conv_sc_floats(32, 3) == 9 * 32 * 8 == L.read_conv_sc_floats(32, 4) and L.read_conv_sc_floats(32, 1) == 9 * 32 * 8
    assert L.read_conv_sc_floats(32, 5) == 0 and L.read_conv_sc_floats(64, 3) == 0 and L.read_conv_sc_floats(16, 3) == 0
    for cout in (1, 3, 4):
        wf = rng.standard_normal((cout, 32, 3, 3)).astype(np.float32)
        wm = rng.standard_normal((cout, 32, 3, 3)).astype(np.float32)
        out = np.full(9 * 32 * 8, np.nan, np.float32)
        _lib.check(L.read_conv_pack_sc_host(32, cout, wf.ctypes.data, wm.ctypes.data, out.ctypes.data))
        o = out.reshape(9, 32, 2, 4)
        want = np.zeros((9, 32, 2, 4), np.float32)
        want[:, :, 0, :cout] = wf.reshape(cout, 32, 9).transpose(2, 1, 0)
        want[:, :, 1, :cout] = wm.reshape(cout, 32, 9).transpose(2, 1, 0)
        assert np.array_equal(o, want)
    assert L.read_conv_pack_sc_host(32, 5, None, None, None) == -22
    assert L.read_conv_pack_sc_host(32, 3, None, None, None) == -22


def test_wgrad_kernel_choice_and_scratch():
    """read_conv_wgrad_family: 4 = summed in the Winograd F(4x4,3x3) domain (3x3 / stride 1, whole 32-channel tiles of input
    channels, whole 4 x 4 pixel tiles), 0 = direct MFMA kernel; the scratch size covers whichever the launch takes; the knob."""
    L = _lib.lib()
    assert L.read_conv_wgrad_family(32, 3, 1, 2176, 256) == 4 and L.read_conv_wgrad_family(256, 3, 1, 272, 32) == 4
    for (cin, k, s_, H, W) in ((8, 3, 1, 64, 64), (48, 3, 1, 64, 64), (32, 3, 2, 64, 64), (32, 1, 1, 64, 64), (32, 4, 2, 64, 64),
                               (32, 3, 1, 62, 64), (32, 3, 1, 64, 66)):
        assert L.read_conv_wgrad_family(cin, k, s_, H, W) == 0, (cin, k, s_, H, W)
    try:
        _lib.check(L.read_tuning_set(b"wgrad_wino", 0))
        assert L.read_conv_wgrad_family(32, 3, 1, 2176, 256) == 0
        v = C.c_int(-1)
        _lib.check(L.read_tuning_get(b"wgrad_wino", C.byref(v)))
        assert v.value == 0
    finally:
        _lib.check(L.read_tuning_set(b"wgrad_wino", 1))
    for (cin, cout, H) in ((32, 32, 2176), (256, 256, 272), (64, 3, 128), (128, 40, 36)):
        tiles = (cin // 32) * ((2 * ((cout + 7) // 8 * 8) + 31) // 32)
        splits_max = -(-256 // tiles)
        assert L.read_conv_wgrad_scratch_floats(cin, cout, 3, H) >= min(splits_max, H // 4) * tiles * 36 * 1024 * 0.5
        assert L.read_conv_wgrad_scratch_floats(cin, cout, 3, H) > 0 and L.read_conv_wgrad_scratch_floats(cin, cout, 3, H + 1) > 0


def test_param_packing_folds_batchnorm():
    L = _lib.lib()
    rng = np.random.default_rng(1)
    cout = 40
    bf, bm, g, b, m = (rng.standard_normal(cout).astype(np.float32) for _ in range(5))
    v = rng.uniform(0.5, 1.5, cout).astype(np.float32)
    out = np.empty(L.read_conv_param_floats(cout), np.float32)
    _lib.check(L.read_conv_pack_params_host(cout, bf.ctypes.data, bm.ctypes.data, g.ctypes.data, b.ctypes.data,
                                            m.ctypes.data, v.ctypes.data, 1e-5, out.ctypes.data))
    o = out.reshape(4, 64)
    scale = g / np.sqrt(v + np.float32(1e-5))
    np.testing.assert_allclose(o[0, :cout], bf)
    np.testing.assert_allclose(o[1, :cout], bm)
    np.testing.assert_allclose(o[2, :cout], scale, rtol=1e-6)
    np.testing.assert_allclose(o[3, :cout], b - m * scale, rtol=1e-5, atol=1e-6)
    assert not o[:, cout:].any()


def test_unet_blob_sizes_and_plan_flops():
    """The launch plan can be built without a GPU (create only records pointers).  As the reference wires the network
    (read_tuning_set("unet_aff_split", 0)): 99 convs + 3 upsamples, algorithmic FLOPs as measured on the reference module
    (BASELINE.md §2).  Default plan: the AFF inputs of coarser levels are multiplied at their own level (3 extra
    launches, 46.0 -> 11.2 GFLOP at 1216x352)."""
    L = _lib.lib()
    raw = sum(2 * (co * ci * k * k + co) + 4 * co for (_, ci, co, k) in UNET_SPEC)
    assert L.read_unet_raw_floats() == raw

    def aff_gflop(H, W, split):
        px = [H * W >> (2 * l) for l in range(4)]
        if not split:
            return sum(4.0 * px[l] * 480 * (32 << l) for l in range(3)) / 1e9
        q = 4.0 * (px[3] * 256 * 224 + px[2] * 128 * 96 + px[1] * 64 * 32)
        r = 4.0 * (px[0] * 32 * 32 + px[1] * 96 * 64 + px[2] * 224 * 128)
        return (q + r) / 1e9

    def up_gflop(H, W, fold):
        # Convs.k over cat[Upsample4(fe_k), r_k], C = 128 / 64 / 32 at levels 2 / 1 / 0: folded, the up-sampled half is applied to
        # fe_k at ITS level (two levels coarser, 1/16 of the pixels) — read_conv_desc.pre_bilinear
        px = [H * W >> (2 * l) for l in range(5)]
        tot = 0.0
        for k, (lvl, Cc) in enumerate(((2, 128), (1, 64), (0, 32))):
            tot += 4.0 * px[lvl] * (2 * Cc) * Cc if not fold else 4.0 * (px[lvl] + px[lvl + 2]) * Cc * Cc
        return tot / 1e9

    for split, fold, launches in ((0, 1, 102), (1, 1, 105), (0, 0, 102), (1, 0, 105)):      # ends on the defaults (1, 0)
        _lib.check(L.read_tuning_set(b"unet_aff_split", split))
        _lib.check(L.read_tuning_set(b"unet_up_fold", fold))
        for (H, W, gflop) in [(352, 1216, 1221.73), (256, 256, 187.06)]:
            need = L.read_unet_workspace_bytes(H, W)
            ws = np.empty(need + 256, np.uint8)
            base = (ws.ctypes.data + 255) // 256 * 256
            pk = np.empty(64, np.float32)
            h = C.c_void_p()
            _lib.check(L.read_unet_create(C.byref(h), (pk.ctypes.data + 15) // 16 * 16, H, W, base, need))
            n = L.read_unet_launch_count(h)
            assert n == launches
            tot = 0.0
            for i in range(n):
                fl = C.c_double()
                L.read_unet_launch_info(h, i, C.byref(fl), None, None, None, None, None)
                tot += fl.value
            want = gflop - aff_gflop(H, W, 0) + aff_gflop(H, W, split) - up_gflop(H, W, 0) + up_gflop(H, W, fold)
            assert abs(tot / 1e9 - want) < 0.01, (split, tot / 1e9, want)
            L.read_unet_destroy(h)
    assert kc_for([8, 56]) == 8 and kc_for([32, 64, 128, 256]) == 16


def test_product_path_refuses_to_run_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from read_amd.raster import PointCloudRasterizer
    with pytest.raises(_lib.ReadHipError):
        PointCloudRasterizer(np.zeros((4, 3), np.float32))
    from read_amd.unet import UNet
    net = UNet().eval()
    with pytest.raises(_lib.ReadHipError), torch.no_grad():
        net(*[torch.zeros(1, 8, 16 >> l, 16 >> l) for l in range(4)])
    with pytest.raises(_lib.ReadHipError):                        # .train() (batch-statistics BatchNorm) is a HIP path too
        net.train()(*[torch.zeros(1, 8, 16 >> l, 16 >> l) for l in range(4)])


def test_no_oracle_import_in_product():
    """read_amd/ must never import oracle/ (the checker is not the product)."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "read_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f


def test_release_library_has_no_debug_entry_points():
    """the measurement probes (csrc/probe.hip) and the kernel timeline switch are entry points of the debug build
    only (include/read_hip_debug.h, libreadhip_debug.so); the product exports none of them."""
    L = C.CDLL(_lib.LIB_PATH if not os.environ.get("READ_HIP_DEBUG") else os.path.join(ROOT, "read_amd", "libreadhip.so"))
    txt = open(os.path.join(ROOT, "include", "read_hip_debug.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    names = sorted(set(re.findall(r"\b(read_debug_[a-z0-9_]+)\s*\(", txt)))
    assert names == sorted(_lib.DEBUG_SIGNATURES) and len(names) == 6
    for n in names:
        assert not hasattr(L, n), f"{n} is exported by the release library"
    assert not any(n.startswith("read_debug") for n in _declared_symbols())
