"""CPU: host-side mirrors of the reference interface (no GPU work): camera conventions, module tree /
state-dict names, input-format parsing, checkpoint round trip, NetAndTexture bookkeeping."""
import os

import numpy as np
import pytest
import torch

from read_amd import camera, synthetic
from read_amd.net_texture import NetAndTexture
from read_amd.pipeline import TexturePipeline, load_model_checkpoint, save_model
from read_amd.render import Scene, parse_input_string
from read_amd.texture import PointTexture
from read_amd.unet import UNet, pack_state, raw_blob_from_state
from tests.unet_spec import UNET_SPEC


def test_proj_matrix_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "proj_1216x352.npz"))
    P = camera.get_proj_matrix(g["K"], (1216, 352), float(g["znear"]), float(g["zfar"]))
    assert np.array_equal(P, g["P"])
    # clip-space convention: a point straight ahead (-z) lands in the image centre, depth in (0,1)
    M = camera.total_matrix(P.astype(np.float32), np.eye(4, dtype=np.float32))[0]
    c = M @ np.array([0, 0, -10, 1], np.float32)
    assert abs(c[0] / c[3]) < 1e-6 and abs(c[1] / c[3]) < 1e-6 and -1 < c[2] / c[3] < 1


def test_level_sizes():
    assert camera.level_sizes(1216, 352) == [(1216, 352), (608, 176), (304, 88), (152, 44), (76, 22)]
    assert camera.level_sizes(250, 130, 3) == [(250, 130), (125, 65), (62, 32)]


def test_unet_state_dict_names_and_blob():
    net = UNet()
    sd = net.state_dict()
    assert len(sd) == 909
    for (path, cin, cout, k) in UNET_SPEC:
        assert tuple(sd[f"{path}.block.conv_f.weight"].shape) == (cout, cin, k, k)
        assert tuple(sd[f"{path}.block.norm.running_var"].shape) == (cout,)
    assert sum(p.numel() for p in net.parameters()) == 30_193_988
    state = synthetic.make_unet_state(UNET_SPEC, 3)
    net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in state.items()})
    assert np.array_equal(raw_blob_from_state(net.state_dict()), raw_blob_from_state(state))
    packed = pack_state(state)
    assert packed.dtype == np.float32 and np.isfinite(packed).all()
    with pytest.raises(ValueError):
        UNet(num_input_channels=3)


def test_aff_weight_blocks_in_the_packed_blob():
    """read_unet_pack_host appends, after the 104 layers, the weight blocks of the split AFF plan (csrc/unet.cpp DerivedInfo):
    columns [ci0, ci0 + cin) of the AFF first convs' (cout, 480) matrices, output channels of the listed AFFs stacked, in the
    usual 1x1 fragment order; the linear partial-sum layers carry zero biases."""
    from read_amd import _lib
    L = _lib.lib()
    state = synthetic.make_unet_state(UNET_SPEC, 11)
    packed = pack_state(state)
    off = 0
    for (path, cin, cout, k) in UNET_SPEC:                                 # the regular layers come first, in table order
        off += L.read_conv_packed_floats(cin, cout, k) + L.read_conv_param_floats(cout)
        if k == 3 and cin % 16 == 0 and not path.startswith("feat_extract.1") and not path.startswith("feat_extract.2") \
                and not path.startswith("feat_extract.6"):
            off += 2 * L.read_conv_wino_floats(cin, cout)                  # 3x3 / stride-1 layers carry G g G^T as well, in the
            if cin >= 32 and cout % 32 == 0:                               # orders of both F(2x2) kernels, and the F(4x4) fragments
                off += L.read_conv_w4_floats(cin, cout)
                if cin % 32 == 0 and not path.startswith("FAM"):           # round 6: ... and their split into f16 piece pairs (not FAM's x1 * x2 layers)
                    off += L.read_conv_w4h_floats(cin, cout)
                if cin % 32 == 0:                                          # ... and the plain weights as f16 piece pairs (the direct kernel; FAM runs on it)
                    off += L.read_conv_d3h_floats(cin, cout)
        if path in ("feat_extract.1", "feat_extract.2", "feat_extract.6", "feat_extract.3", "feat_extract.4", "feat_extract.7"):   # the stride-2 layers
            off += L.read_conv_dkh_floats(cin, cout, k)                    # (3x3 and 4x4): the direct split-operand kernel's operand
        if k == 1 and cin <= 256:                                          # 1x1 layers: the split-operand pixel-lane kernel's operand (f16 piece pairs)
            off += L.read_conv_dkh_floats(cin, cout, 1)
        if k == 3 and cin == 8:                                            # the layers over the 8-channel pyramid: implicit-GEMM operand (f16 piece pairs)
            off += L.read_conv_t3h_floats(cin, cout)
        if L.read_conv_sc_floats(cin, cout) and k == 3:                    # the 32 -> 3 layer: the vector-pipe order, 64-byte aligned
            off = (off + 15) // 16 * 16 + L.read_conv_sc_floats(cin, cout)
    aff = lambda *ks: tuple(f"AFFs.{k}.conv.0" for k in ks)                # noqa: E731
    derived = [("AFFq3", 224, 256, aff(0, 1, 2)), ("AFFq2", 96, 128, aff(0, 1)), ("AFFq1", 32, 64, aff(0)),
               ("AFFs.0.conv.0r", 0, 32, aff(0)), ("AFFs.1.conv.0r", 0, 96, aff(1)), ("AFFs.2.conv.0r", 0, 224, aff(2))]
    # round 5: Convs.k over cat[Upsample4(fe), r] split into the half applied at fe's level (.u) and the half at r's level (.r)
    for k, c in enumerate((128, 64, 32)):
        derived += [(f"Convs.{k}.u", 0, c, (f"Convs.{k}",)), (f"Convs.{k}.r", c, c, (f"Convs.{k}",))]
    rng = np.random.default_rng(5)
    for name, ci0, cin, affs in derived:
        wf = np.concatenate([np.asarray(state[f"{a}.block.conv_f.weight"])[:, ci0:ci0 + cin, 0, 0] for a in affs])
        wm = np.concatenate([np.asarray(state[f"{a}.block.conv_m.weight"])[:, ci0:ci0 + cin, 0, 0] for a in affs])
        cout = wf.shape[0]
        n = L.read_conv_packed_floats(cin, cout, 1)
        blk = packed[off:off + n].reshape(cin // 8, (cout + 31) // 32 * 2, 64, 4)          # [k8 step][tile (f, m)][lane][4]
        for _ in range(300):
            s_, nt, lane, j = (int(rng.integers(0, d)) for d in blk.shape)
            co, ci = (nt // 2) * 32 + lane % 32, 8 * s_ + 4 * (lane // 32) + j
            want = (wm if nt % 2 else wf)[co, ci] if co < cout else 0.0
            assert blk[s_, nt, lane, j] == want, (name, s_, nt, lane, j)
        par = packed[off + n:off + n + L.read_conv_param_floats(cout)].reshape(4, -1)
        assert not par[0].any() and not par[1].any()                         # zero biases (the gated finals use their AFF's own block)
        off += n + L.read_conv_param_floats(cout)
        # round 6: ... followed by the same block as f16 piece pairs (read_conv_pack_dkh_host, ksize 1; tests/d3h_ref.py restates it)
        from tests.d3h_ref import pack_d1h_blob
        nh = L.read_conv_dkh_floats(cin, cout, 1)
        assert nh == cin * 2 * ((cout + 31) // 32 * 32) + 2 * ((cout + 31) // 32 * 32)
        assert np.array_equal(packed[off:off + nh].view(np.uint32), pack_d1h_blob(wf, wm).view(np.uint32)), name
        off += nh
    assert off == L.read_unet_packed_floats()


def test_input_format_tokens(golden_dir):
    """The whole DSL against the reference's own parse_input_string (tests/golden/make_tokens_golden.py executes its
    source text): every key and value equal."""
    import json
    from read_amd.render import is_point_id_pyramid
    g = json.load(open(os.path.join(golden_dir, "input_tokens.json")))
    assert len(g["tokens"]) >= 19
    for tok, want in g["tokens"].items():
        got = parse_input_string(tok)
        got["mode"] = list(got["mode"])
        assert got == want, (tok, got, want)
    for bad in g["value_error"]:
        with pytest.raises(ValueError):
            parse_input_string(bad)
    # the other direction (READ/gl/dataset.py:85-122): the reference's generate_input_string on every parsed configuration and
    # on the configurations of its own test_generate_parse (:124-200)
    from read_amd.render import generate_input_string
    for tok, want in g["generated"].items():
        got = generate_input_string(parse_input_string(tok))
        if tok.startswith("labels"):
            assert want == "_p1" and got == tok      # the reference has no branch for MODE_LABEL; here the round trip holds
        else:
            assert got == want == tok, (tok, got, want)
    assert len(g["reference_test_generate_parse"]) == 6
    for case in g["reference_test_generate_parse"]:
        cfg = dict(case["config"], mode=tuple(case["config"]["mode"]))
        assert generate_input_string(cfg) == case["string"], case
        assert parse_input_string(case["string"]) == cfg
    with pytest.raises(ValueError):
        generate_input_string({"mode": (3, 7), "draw_points": False})
    assert is_point_id_pyramid("uv_1d_p1, uv_1d_p1_ds1, uv_1d_p1_ds2, uv_1d_p1_ds3, uv_1d_p1_ds4")
    for fmt in ("uv_1d_p1, uv_1d_p2_ds1", "uv_1d_p1_ds1", "colors_p1", "uv_1d_ps4", "uv_1d_p1, uv_1d_p1_ds2"):
        assert not is_point_id_pyramid(fmt)


def test_scene_setters_and_refusals():
    """NNScene's setter surface (READ/gl/programs.py:300-415) on the GL-free Scene."""
    xyz = synthetic.make_cloud(50)
    s = Scene()
    s.set_vertices(xyz, colors=np.ones((50, 3)), normals=np.zeros((50, 3)), uv1d=np.arange(50), uv2d=np.zeros((50, 2)))
    assert not s.augmented()
    s.set_point_discard(np.arange(50) % 2 == 0)
    assert s.augmented() and s.point_discard.dtype == bool
    s.set_point_discard(None)
    s.set_point_perturb(np.zeros((50, 2)))
    assert s.augmented()
    s.set_point_perturb(None)
    s.set_point_drop(0.25, seed=3)
    s.set_point_perturb_seeded(0.1, seed=4)
    assert s.point_drop == (0.25, 3) and s.point_perturb_seeded == (0.1, 4)
    s.set_params(**parse_input_string("colors_ps7_ds1"))
    assert s.params["point_size"] == 7 and s.params["splat_mode"] and s.params["mode"] == (0, None)
    s.set_point_sizes(np.full(50, 3.0))                    # per-point sizes (programs.py:339-345): accepted, make the scene "augmented"
    assert s.point_sizes.dtype == np.float32 and s.augmented()
    with pytest.raises(ValueError):
        s.set_point_sizes(np.ones(49))
    with pytest.raises(NotImplementedError):
        s.set_vertices(xyz, uv1d=np.arange(50)[::-1])
    with pytest.raises(AssertionError):
        s.set_vertices(xyz, colors=np.ones((49, 3)))


def test_scene_total_matrix_matches_myrender_formula():
    s = Scene(synthetic.make_cloud(10))
    proj, pose = synthetic.make_proj(64, 48, f=40.0), synthetic.sweep_pose(5)
    s.set_proj_matrix(proj)
    s.set_camera_view(pose)
    np.testing.assert_allclose(s.total_matrix()[0], camera.total_matrix(proj, pose)[0], rtol=1e-6, atol=1e-6)


def test_checkpoint_roundtrip(tmp_path):
    t = PointTexture(8, 100, init_method='rand')
    p = tmp_path / "PointTexture_x.pth"
    save_model(str(p), t, args={'descriptor_size': 8})
    ck = torch.load(p, weights_only=False)
    assert set(ck) == {'state_dict', 'args'} and list(ck['state_dict']) == ['texture_']
    assert tuple(ck['state_dict']['texture_'].shape) == (1, 8, 100)
    t2 = load_model_checkpoint(str(p), PointTexture(8, 100))
    assert torch.equal(t2.texture_, t.texture_)


def test_pipeline_inference_create_and_netandtexture_bookkeeping():
    pl = TexturePipeline()
    pl.create({'inference': True, 'n_points': 64, 'descriptor_size': 8, 'texture_activation': 'none',
               'texture_ckpt': None, 'use_mesh': False})
    for attr in ('model', 'net', 'textures', 'args'):
        assert hasattr(pl, attr)
    assert isinstance(pl.model, NetAndTexture) and pl.get_net() is pl.net
    m = pl.model
    assert '0' not in m._modules
    m.load_textures(0)
    assert m._modules['0'] is pl.textures[0]
    assert any(k == '0.texture_' for k in m.state_dict())
    m.unload_textures()
    assert '0' not in m._modules
    # the reference's dotted path resolves through the alias package
    from READ.pipelines.ogl import TexturePipeline as Alias
    assert Alias is TexturePipeline
    import pcpr
    assert callable(pcpr.forward)
    with pytest.raises(RuntimeError):
        pcpr.forward(torch.zeros(4, 3), torch.zeros(4, 4), 8, 8, 512)     # "batch_size check": total_m must be 3-D
    with pytest.raises(RuntimeError):
        pcpr.forward(torch.zeros(4, 3, dtype=torch.float64), torch.zeros(1, 4, 4), 8, 8, 512)


def test_pipeline_optimizer_is_adam_on_the_cpu():
    """TexturePipeline's optimizer (read_amd.pipeline._DeviceAdam) only switches to torch's fused kernel when every parameter is
    on the GPU at its first step; on the CPU it is torch.optim.Adam step for step (same state_dict keys, same trajectory)."""
    from read_amd.pipeline import _DeviceAdam
    torch.manual_seed(3)
    a = [torch.nn.Parameter(torch.randn(7)), torch.nn.Parameter(torch.randn(3, 4))]
    b = [torch.nn.Parameter(p.detach().clone()) for p in a]
    oa, ob = _DeviceAdam(a, lr=1e-2), torch.optim.Adam(b, lr=1e-2)
    for _ in range(3):
        for p, q in zip(a, b):
            g = torch.randn_like(p)
            p.grad, q.grad = g.clone(), g.clone()
        oa.step()
        ob.step()
    assert not oa.param_groups[0].get('fused')
    for p, q in zip(a, b):
        assert torch.equal(p, q)
    assert oa.state_dict()['param_groups'][0].keys() == ob.state_dict()['param_groups'][0].keys()


def test_lean_packed_layout_is_a_subset_of_the_full_one():
    """a blob that carries only the fragment orders the default plan reads (READ_UNET_LAYOUT_LEAN): less than half
    of the full blob's bytes, every value of it present in the full blob (same packers, fewer orders), layouts told apart by length."""
    from read_amd import _lib
    from read_amd.unet import LAYOUT_FULL, LAYOUT_LEAN, layout_of
    L = _lib.lib()
    n_full, n_lean = L.read_unet_packed_floats_layout(LAYOUT_FULL), L.read_unet_packed_floats_layout(LAYOUT_LEAN)
    assert n_full == L.read_unet_packed_floats() and 0 < n_lean < 0.5 * n_full and L.read_unet_packed_floats_layout(7) == 0
    state = synthetic.make_unet_state(UNET_SPEC, 3)
    full, lean = pack_state(state, layout=LAYOUT_FULL), pack_state(state, layout=LAYOUT_LEAN)
    assert full.size == n_full and lean.size == n_lean and np.isfinite(lean).all()
    assert layout_of(full) == LAYOUT_FULL and layout_of(lean) == LAYOUT_LEAN
    with pytest.raises(_lib.ReadHipError):
        layout_of(np.zeros(12345, np.float32))
    # the F(4x4) order of one layer, found by value in both blobs
    w4 = np.empty(L.read_conv_w4_floats(64, 64), np.float32)
    wf = np.ascontiguousarray(state["Encoder.1.layers.0.main.0.block.conv_f.weight"], np.float32)
    wm = np.ascontiguousarray(state["Encoder.1.layers.0.main.0.block.conv_m.weight"], np.float32)
    _lib.check(L.read_conv_pack_w4_host(64, 64, wf.ctypes.data, wm.ctypes.data, w4.ctypes.data))
    probe = w4[:64].tobytes()
    # round 6: the lean blob carries that layer's SPLIT operand (f16 piece pairs, the kernel the default plan runs) instead of the
    # fp32 order, the full blob both; FAM's x1 * x2 layers stay on the fp32 kernel and keep the fp32 order in the lean blob too
    w4h = np.empty(L.read_conv_w4h_floats(64, 64), np.float32)
    _lib.check(L.read_conv_pack_w4h_host(64, 64, wf.ctypes.data, wm.ctypes.data, w4h.ctypes.data))
    probe_h = w4h[:64].tobytes()
    lean_b, full_b = lean.tobytes(), full.tobytes()
    assert probe_h in lean_b and probe_h in full_b and probe in full_b and probe not in lean_b
    ff = np.ascontiguousarray(state["FAM2.merge.block.conv_f.weight"], np.float32)
    fm = np.ascontiguousarray(state["FAM2.merge.block.conv_m.weight"], np.float32)
    d3f = np.empty(L.read_conv_d3h_floats(64, 64), np.float32)               # FAM: the direct split-operand kernel's operand, in both blobs
    _lib.check(L.read_conv_pack_d3h_host(64, 64, ff.ctypes.data, fm.ctypes.data, d3f.ctypes.data))
    assert d3f[:64].tobytes() in lean_b and d3f[:64].tobytes() in full_b
    w4f = np.empty(L.read_conv_w4_floats(64, 64), np.float32)
    _lib.check(L.read_conv_pack_w4_host(64, 64, ff.ctypes.data, fm.ctypes.data, w4f.ctypes.data))
    assert w4f[:64].tobytes() in full_b and w4f[:64].tobytes() not in lean_b
