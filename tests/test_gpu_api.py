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
    assert fast.shape == (H, W, 4) and bool((fast[..., 3] == 1).all())
    slow = ogl.infer({k: v.permute(2, 0, 1)[None] for k, v in maps.items()})['output']
    torch.testing.assert_close(fast, slow, rtol=0, atol=1e-6)
    with torch.no_grad():
        ref = unet_torch.net_and_texture_forward(state, tex.texture_.detach().cpu().numpy(), oi)[0]
    assert unet_torch.psnr(fast[..., :3].permute(2, 0, 1).cpu(), ref) >= 80.0
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
            assert unet_torch.psnr(out[b].cpu(), ref) >= 80.0
