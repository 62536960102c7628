"""GPU: the reference-shaped Python interfaces (B1-B6 of SURVEY.md §8b) end to end, against the oracle."""
import types

import numpy as np
import pytest
import torch

import oracle
from oracle import unet_torch
from read_amd import camera, pcpr, synthetic
from read_amd.net_texture import NetAndTexture
from read_amd.ogl import OGL
from read_amd.render import MultiscaleRender, MyRender, Scene
from read_amd.texture import PointTexture
from read_amd.unet import UNet
from tests.unet_spec import UNET_SPEC

pytestmark = pytest.mark.gpu

FMT = "uv_1d_p1, uv_1d_p1_ds1, uv_1d_p1_ds2, uv_1d_p1_ds3, uv_1d_p1_ds4"


def test_pcpr_forward_contract(hip):
    """B1: same signature/returns as the reference extension (CPU float tensors, float ids)."""
    W, H, N = 96, 64, 20_000
    pts = torch.from_numpy(synthetic.make_cloud(N))
    tm = torch.from_numpy(camera.total_matrix(synthetic.make_proj(W, H, f=70.0),
                                              np.stack([synthetic.sweep_pose(1), synthetic.sweep_pose(9)])))
    index, depth = pcpr.forward(pts, tm, W, H, 512)
    assert index.shape == depth.shape == (2, H, W) and index.dtype == depth.dtype == torch.float32
    assert not index.is_cuda and not depth.is_cuda
    for b in range(2):
        oi, od = oracle.raster_level(pts.numpy(), tm[b].numpy(), W, H)
        assert np.array_equal(index[b].numpy(), oracle.index_to_float(oi))
        assert np.array_equal(depth[b].numpy().view(np.uint32), od.view(np.uint32))
    index2, _ = pcpr.forward(pts, tm, W, H, 512)          # cached cloud, same answer
    assert torch.equal(index, index2)


def _fake_ds(xyz, W, H, ds_id=0):
    return types.SimpleNamespace(id=ds_id, tgt_sh=(W, H), input_format=FMT,
                                 scene_data={'pointcloud': {'xyz': xyz}})


def test_myrender_contract(hip):
    """B2: out_dict / depth_dict keyed by input_format tokens, (B,1,h,w) float32 CPU tensors."""
    W, H = 128, 96
    xyz = synthetic.make_cloud(30_000)
    r = MyRender([_fake_ds(xyz, W, H)])
    proj = np.stack([synthetic.make_proj(W, H, f=90.0)] * 2)
    view = np.stack([synthetic.sweep_pose(2), synthetic.sweep_pose(30)])
    data = {'input': {'id': torch.tensor([0, 0])}, 'proj_matrix': torch.from_numpy(proj),
            'view_matrix': torch.from_numpy(view)}
    out, dep = r.render(data)
    keys = FMT.replace(' ', '').split(',')
    assert list(out) == ['id'] + keys and list(dep) == keys
    tm = camera.total_matrix(proj, view)
    for b in range(2):
        oi, od = oracle.raster_multiscale(xyz, tm[b], W, H, 5)
        for l, k in enumerate(keys):
            assert out[k].shape == (2, 1, H >> l, W >> l) and not out[k].is_cuda
            assert np.array_equal(out[k][b, 0].numpy(), oracle.index_to_float(oi[l]))
            assert np.array_equal(dep[k][b, 0].numpy().view(np.uint32), od[l].view(np.uint32))


def _model(N, seed=3):
    state = synthetic.make_unet_state(UNET_SPEC, seed)
    net = UNet()
    net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in state.items()})
    tex = PointTexture(8, N, init_method='rand')
    model = NetAndTexture(net, {0: tex})
    model.load_textures(0)
    return model.cuda().eval(), state, tex


def test_multiscale_render_and_ogl_infer(hip):
    """B3 + B4: scene camera -> RGBA frame; fast path == dict path == oracle."""
    W, H, N = 128, 64, 25_000
    xyz = synthetic.make_cloud(N)
    model, state, tex = _model(N)
    scene = Scene(xyz)
    proj, pose = synthetic.make_proj(W, H, f=80.0), synthetic.sweep_pose(4)
    scene.set_proj_matrix(proj)
    scene.set_camera_view(pose)
    msr = MultiscaleRender(scene, FMT, (W, H), out_buffer_location='torch')
    maps = msr.render()
    M = camera.total_matrix(proj, pose)[0]
    oi, _ = oracle.raster_multiscale(xyz, M, W, H, 5)
    for l, k in enumerate(FMT.replace(' ', '').split(',')):
        assert maps[k].shape == (H >> l, W >> l, 3)
        assert np.array_equal(maps[k][..., 0].cpu().numpy(), oracle.index_to_float(oi[l]))
        assert float(maps[k][..., 1:].abs().sum()) == 0.0
    flipped = MultiscaleRender(scene, FMT, (W, H), out_buffer_location='torch', gl_frame=True).render()
    assert torch.equal(flipped['uv_1d_p1'], maps['uv_1d_p1'].flip([0]))

    ogl = OGL.from_model(scene, model, FMT, (W, H))
    fast = ogl.infer()['output']
    assert ogl.last_path == 'fast'                                         # the device-resident branch really ran
    assert fast.shape == (H, W, 4) and bool((fast[..., 3] == 1).all())
    slow = ogl.infer({k: v.permute(2, 0, 1)[None] for k, v in maps.items()})['output']
    assert ogl.last_path == 'dict'
    torch.testing.assert_close(fast, slow, rtol=0, atol=1e-6)
    with torch.no_grad():
        ref = unet_torch.net_and_texture_forward(state, tex.texture_.detach().cpu().numpy(), oi)[0]
    assert unet_torch.psnr(fast[..., :3].permute(2, 0, 1).cpu(), ref) >= 120.0
    with pytest.raises(AssertionError):
        OGL.from_model(scene, model, FMT, (100, 64))                       # viewport must be a multiple of 16


def test_netandtexture_forward_contract(hip):
    """B6: batch of 2 items, dict input loses 'id', output (B,3,H,W)."""
    W, H, N = 64, 48, 6_000
    model, state, tex = _model(N, seed=5)
    rng = np.random.default_rng(0)
    keys = FMT.replace(' ', '').split(',')
    inputs = {'id': torch.tensor([0, 0])}
    maps = []
    for l, k in enumerate(keys):
        m = rng.integers(0, N, (2, 1, H >> l, W >> l))
        maps.append(m)
        inputs[k] = torch.from_numpy(m).float().cuda()
    with torch.no_grad():
        out = model(inputs)
    assert 'id' not in inputs and out.shape == (2, 3, H, W)
    with torch.no_grad():
        for b in range(2):
            ref = unet_torch.net_and_texture_forward(state, tex.texture_.detach().cpu().numpy(),
                                                     [m[b, 0].astype(np.int32) for m in maps])[0]
            assert unet_torch.psnr(out[b].cpu(), ref) >= 120.0


def test_scene_directory_and_checkpoints_to_frame(hip, tmp_path):
    """SURVEY §8f rows 1-2 end to end: a scene directory on disk (scene.yaml + pointcloud.ply + Metashape camera.xml) and
    reference-format checkpoints ({'state_dict', 'args'}) -> load_scene_data / setup_scene -> OGL(...) -> RGBA frame,
    equal to the oracle run on the same files' contents."""
    from read_amd import scene_io
    from read_amd.pipeline import save_model
    W, H, N = 128, 64, 20_000
    xyz = synthetic.make_cloud(N, seed=5)
    scene_io.write_ply(str(tmp_path / "pointcloud.ply"), xyz, rgb=np.zeros((N, 3), np.uint8), normals=np.zeros((N, 3)))
    poses = [synthetic.sweep_pose(3), synthetic.sweep_pose(9)]
    cams = []
    for k, p in enumerate(poses):
        m = p.astype(np.float64).copy()
        m[:, 1:3] *= -1                                   # Metashape convention on disk; the loader flips it back
        cams.append(f'<camera id="{k}" label="{k}"><transform>' + " ".join(repr(float(v)) for v in m.reshape(-1))
                    + "</transform></camera>")
    (tmp_path / "camera.xml").write_text(
        f'<document><chunk><sensors><sensor id="0"><calibration><resolution width="{W}" height="{H}"/>'
        f'<f>80.0</f></calibration></sensor></sensors><cameras>{"".join(cams)}</cameras></chunk></document>')
    ck = tmp_path / "run" / "checkpoints"
    ck.mkdir(parents=True)
    state = synthetic.make_unet_state(UNET_SPEC, 21)
    net = UNet()
    net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in state.items()})
    tex = PointTexture(8, N, init_method='rand')
    args = dict(input_format=FMT, descriptor_size=8, texture_activation='none', n_points=N, supersampling=1,
                pipeline='READ.pipelines.ogl.TexturePipeline', inference=False, lr=1e-4, texture_lr=1e-1)
    save_model(str(ck / "UNet_stage_0_epoch_1_net.pth"), net, args=args)
    save_model(str(ck / "PointTexture_stage_0_epoch_1.pth"), tex, args=args)
    (tmp_path / "scene.yaml").write_text(
        f"viewport_size: [{W}, {H}]\nintrinsic_matrix: camera.xml\nview_matrix: camera.xml\npointcloud: pointcloud.ply\n"
        f"net_path: {tmp_path / 'run'}\nckpt: UNet_stage_0_epoch_1_net.pth\ntexture_ckpt: PointTexture_stage_0_epoch_1.pth\n")

    sd = scene_io.load_scene_data(str(tmp_path / "scene.yaml"))
    assert sd["camera_labels"] == ["0", "1"] and np.allclose(sd["view_matrix"][1], poses[1], atol=1e-6)
    scene = Scene()
    scene_io.setup_scene(scene, sd)
    K = sd["intrinsic_matrix"]
    proj = camera.get_proj_matrix(K, sd["config"]["viewport_size"], 0.1, 1000.).astype(np.float32)
    scene.set_proj_matrix(proj)
    ogl = OGL(scene, sd, sd["config"]["viewport_size"], sd["net_ckpt"], sd["tex_ckpt"], out_buffer_location='torch')
    for k in (1, 0):
        scene.set_camera_view(sd["view_matrix"][k])
        out = ogl.infer()["output"]
        assert out.shape == (H, W, 4) and bool((out[..., 3] == 1).all())
        M = camera.total_matrix(proj, np.asarray(sd["view_matrix"][k], np.float32))[0]
        oi, _ = oracle.raster_multiscale(np.asarray(sd["pointcloud"]["xyz"], np.float32), M, W, H, 5)
        with torch.no_grad():
            ref = unet_torch.net_and_texture_forward(state, tex.texture_.detach().cpu().numpy(), oi)[0]
        assert unet_torch.psnr(out[..., :3].permute(2, 0, 1).cpu(), ref) >= 120.0, f"camera {k}"


def test_out_of_range_ids_raise_without_a_sync_per_frame(hip):
    """NetAndTexture's all-uv path checks the point ids of a lookup asynchronously (the reference's index_select reports them
    through an asynchronous device assert): the IndexError surfaces at the next lookup or at check_ids()."""
    from read_amd.net_texture import NetAndTexture
    from read_amd.texture import PointTexture
    W, H, N = 64, 48, 500
    net = UNet()
    tex = PointTexture(8, N, init_method='rand')
    model = NetAndTexture(net, {0: tex})
    model.load_textures(0)
    model.cuda().eval()
    good = {'id': 0}
    bad = {'id': 0}
    for l, k in enumerate(FMT.replace(' ', '').split(',')):
        good[k] = torch.randint(0, N, (1, 1, H >> l, W >> l)).float()
        bad[k] = good[k].clone()
    bad['uv_1d_p1'][0, 0, 3, 5] = float(N + 7)
    with torch.no_grad():
        model(dict(good))
        model.check_ids()                                          # nothing wrong so far
        model(dict(bad))                                           # the gather clamps; the verdict is queued
        with pytest.raises(IndexError):
            model.check_ids()
        model(dict(good))
        model.check_ids()
    # ... and the module-level lookup (the per-item training path: PointTexture.forward) queues its verdict the same way
    ids_ok = torch.randint(0, N, (1, 1, 12, 20), device="cuda").float()
    ids_bad = ids_ok.clone()
    ids_bad[0, 0, 2, 3] = -1.0
    with torch.no_grad():
        tex(ids_ok)
        tex.check_ids()
        tex(ids_bad)
        with pytest.raises(IndexError):
            tex.check_ids()
        tex(ids_ok)
        tex.check_ids()


def test_bilinear_down_equals_torch_interpolate(hip):
    """compose.py:162-163 for network inputs that mix non-uv tokens with texture samples: the HIP node against
    F.interpolate(scale_factor=1/ss, mode='bilinear'), forward and adjoint."""
    import torch.nn.functional as F
    from read_amd.texture import bilinear_down
    rng = np.random.default_rng(9)
    for ss, shape in ((2, (2, 5, 12, 20)), (3, (1, 3, 9, 15)), (4, (1, 11, 16, 8))):
        x = torch.from_numpy(rng.standard_normal(shape).astype(np.float32))
        xr = x.clone().requires_grad_(True)
        yr = F.interpolate(xr, scale_factor=1. / ss, mode='bilinear')
        g = torch.from_numpy(rng.standard_normal(tuple(yr.shape)).astype(np.float32))
        yr.backward(g)
        xd = x.cuda().requires_grad_(True)
        y = bilinear_down(xd, ss)
        torch.testing.assert_close(y.cpu(), yr.detach(), rtol=1e-5, atol=1e-6)
        y.backward(g.cuda())
        torch.testing.assert_close(xd.grad.cpu(), xr.grad, rtol=1e-5, atol=1e-6)
