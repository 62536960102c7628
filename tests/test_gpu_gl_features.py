"""Parity of the GL-twin rasteriser features with their oracle restatement (oracle/raster.c,
oracle_raster_level_gl): point sizes, perspective "ps" splats, point discard / seeded drop, clip-space perturbation,
supersampling, and the colour modes of the input-format DSL.  Index / depth bit-exact; colours exact gathers; the
supersampled descriptor lookup within fp32 round-off of torch's bilinear interpolate (the reference's own op)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import oracle
from oracle import unet_torch
from read_amd import camera, synthetic
from read_amd.raster import PointCloudRasterizer
from read_amd.render import MultiscaleRender, Scene
from read_amd.texture import gather_pyramid, texture_to_rows

pytestmark = pytest.mark.gpu


def _same(got_i, got_d, ref, what):
    oi, od = ref
    gi, gd = got_i[0].cpu().numpy(), got_d[0].cpu().numpy()
    assert np.array_equal(gi, oi), f"{what}: {(gi != oi).sum()} index px differ"
    assert np.array_equal(gd.view(np.uint32), od.view(np.uint32)), f"{what}: depth differs"


def test_point_sizes_and_perspective_splats(hip):
    W, H, N = 304, 176, 200_000
    xyz = synthetic.make_cloud(N, seed=4)
    proj = synthetic.make_proj(W, H, f=180.0)
    M = camera.total_matrix(proj, synthetic.sweep_pose(6))
    r = PointCloudRasterizer(xyz)
    # default options == the 1-px rasteriser
    i1, d1 = r.render_gl(M, W, H)
    _same(i1, d1, oracle.raster_level(xyz, M[0], W, H), "p1 vs 1-px oracle")
    for size in (2, 3, 4, 7):
        i, d = r.render_gl(M, W, H, point_size=size)
        _same(i, d, oracle.raster_level_gl(xyz, M[0], W, H, point_size=size), f"p{size}")
    for size, mn in ((20, 1.0), (60, 1.0), (40, 2.5)):
        i, d = r.render_gl(M, W, H, point_size=size, relative=True, min_point_size=mn)
        _same(i, d, oracle.raster_level_gl(xyz, M[0], W, H, point_size=size, relative=True, min_point_size=mn), f"ps{size}")
    # every level is drawn at its own size with the same pixel size (no pyramid identity for p > 1)
    for l in (1, 2, 3):
        w, h = W >> l, H >> l
        i, d = r.render_gl(M, w, h, point_size=3)
        _same(i, d, oracle.raster_level_gl(xyz, M[0], w, h, point_size=3), f"p3 level {l}")
    # even sizes and "ps" splats that really exceed a pixel; per-point size arrays (NNScene.set_point_sizes, programs.py:183-187)
    for size in (6, 12):
        i, d = r.render_gl(M, W, H, point_size=size)
        _same(i, d, oracle.raster_level_gl(xyz, M[0], W, H, point_size=size), f"p{size}")
    for size in (400, 1500):
        i, d = r.render_gl(M, W, H, point_size=size, relative=True)
        ref = oracle.raster_level_gl(xyz, M[0], W, H, point_size=size, relative=True)
        assert (ref[0] != 0).sum() > 1.3 * (i1[0].cpu().numpy() != 0).sum()      # these splats are larger than one pixel
        _same(i, d, ref, f"ps{size}")
    rng = np.random.default_rng(12)
    sizes = rng.uniform(0.5, 9.0, N).astype(np.float32)
    i, d = r.render_gl(M, W, H, point_sizes=sizes)
    _same(i, d, oracle.raster_level_gl(xyz, M[0], W, H, point_sizes=sizes), "per-point sizes")
    i, d = r.render_gl(M, W, H, point_sizes=200.0 * sizes, relative=True, min_point_size=1.5)
    _same(i, d, oracle.raster_level_gl(xyz, M[0], W, H, point_sizes=200.0 * sizes, relative=True, min_point_size=1.5),
          "per-point sizes, perspective")
    with pytest.raises(ValueError):
        r.render_gl(M, W, H, point_sizes=sizes[:-1])
    # through the scene API: set_point_sizes overrides the token's size for every token (programs.py:404-406)
    scene = Scene(xyz)
    scene.set_proj_matrix(proj)
    scene.set_camera_view(synthetic.sweep_pose(6))
    scene.set_point_sizes(sizes)
    out = MultiscaleRender(scene, "uv_1d_p1, uv_1d_p3_ds1", (W, H), out_buffer_location='torch').render()
    ref0 = oracle.raster_level_gl(xyz, M[0], W, H, point_sizes=sizes)[0]
    ref1 = oracle.raster_level_gl(xyz, M[0], W // 2, H // 2, point_sizes=sizes)[0]
    assert np.array_equal(out["uv_1d_p1"][..., 0].cpu().numpy(), oracle.index_to_float(ref0))
    assert np.array_equal(out["uv_1d_p3_ds1"][..., 0].cpu().numpy(), oracle.index_to_float(ref1))
    # the workspace is left EMPTY: the plain path renders the same frame afterwards
    idx, dep = r.render(M, W, H, 1)
    _same(idx[0], dep[0], oracle.raster_level(xyz, M[0], W, H), "plain after gl")


def test_discard_drop_and_perturb(hip):
    W, H, N = 256, 160, 300_000
    xyz = synthetic.make_cloud(N, seed=9)
    proj = synthetic.make_proj(W, H, f=150.0)
    M = camera.total_matrix(proj, synthetic.sweep_pose(11))
    r = PointCloudRasterizer(xyz)
    rng = np.random.default_rng(0)
    mask = rng.random(N) < 0.3                                   # READ/datasets/dynamic.py:236 with an explicit array
    i, d = r.render_gl(M, W, H, discard=mask)
    _same(i, d, oracle.raster_level_gl(xyz, M[0], W, H, discard=mask), "discard mask")
    assert not np.isin(i[0].cpu().numpy()[d[0].cpu().numpy() > 0], np.nonzero(mask)[0]).any()
    for p, seed in ((0.5, 7), (0.05, 123456789), (1.0, 1)):
        i, d = r.render_gl(M, W, H, drop=(p, seed))
        _same(i, d, oracle.raster_level_gl(xyz, M[0], W, H, drop=(p, seed)), f"seeded drop {p}")
        _same(i, d, oracle.raster_level_gl(xyz, M[0], W, H, discard=oracle.drop_mask(N, p, seed)), f"drop {p} == its mask")
    pert = (0.4 * (rng.random((N, 2)) - 0.5)).astype(np.float32)
    i, d = r.render_gl(M, W, H, perturb=pert)
    _same(i, d, oracle.raster_level_gl(xyz, M[0], W, H, perturb=pert), "perturb array")
    i, d = r.render_gl(M, W, H, perturb_hash=(0.25, 99), point_size=2, drop=(0.2, 5))
    _same(i, d, oracle.raster_level_gl(xyz, M[0], W, H, perturb_hash=(0.25, 99), point_size=2, drop=(0.2, 5)), "all together")
    _same(i, d, oracle.raster_level_gl(xyz, M[0], W, H, perturb=oracle.perturb_array(N, 0.25, 99), point_size=2,
                                       discard=oracle.drop_mask(N, 0.2, 5)), "seeded == explicit arrays")


def test_supersampled_descriptor_lookup(hip):
    """read_gather_forward_ss == PointTexture lookup at ss x + F.interpolate(scale_factor=1/ss, 'bilinear')
    (READ/models/compose.py:162-163), for even and odd factors and every activation."""
    rng = np.random.default_rng(2)
    N, Cc = 5000, 8
    tex = rng.standard_normal((Cc, N)).astype(np.float32)
    rows = texture_to_rows(torch.from_numpy(tex).cuda())
    for ss in (2, 3, 4):
        sizes = [(2, 24, 40), (2, 12, 20), (2, 6, 10)]
        idx = [torch.from_numpy(rng.integers(0, N, (b, h * ss, w * ss)).astype(np.int32)).cuda() for (b, h, w) in sizes]
        for act in ("none", "sigmoid", "tanh"):
            got = gather_pyramid(rows, idx, act, ss=ss)
            for g, i, (b, h, w) in zip(got, idx, sizes):
                assert tuple(g.shape) == (b, h, w, Cc)
                full = torch.from_numpy(tex)[:, i.cpu().long()].permute(1, 0, 2, 3)          # (B,C,ssH,ssW)
                full = torch.sigmoid(full) if act == "sigmoid" else torch.tanh(full) if act == "tanh" else full
                ref = F.interpolate(full, scale_factor=1. / ss, mode='bilinear')
                torch.testing.assert_close(g.permute(0, 3, 1, 2).cpu(), ref, rtol=1e-5, atol=1e-6)


def _colour_oracle(scene, cfg_mode, idx, dep, M, view):
    """NumPy restatement of the vertex colours of READ/gl/programs.py:133-181 for the winning point of every pixel."""
    mode0, mode1 = cfg_mode
    xyz, nrm, rgb = scene.xyz, scene.normals, scene.colors
    covered = (dep != 0) | (idx != 0)
    if mode0 == 0:
        col = rgb[idx]
    elif mode0 == 5:
        col = np.zeros(idx.shape + (3,), np.float32)
        col[..., 0] = nrm[idx][..., 0] / np.float32(255.)
    elif mode0 == 4:
        col = (xyz[idx] - scene.xyz_min) / (scene.xyz_max - scene.xyz_min + np.float32(1e-9))
    elif mode0 == 2:
        p = xyz[idx]
        d = M[2, 0] * p[..., 0] + M[2, 1] * p[..., 1] + M[2, 2] * p[..., 2] + M[2, 3]
        col = np.repeat(d[..., None], 3, -1)
    else:
        cam = view[:3, 3]
        unit = lambda v: v / np.linalg.norm(v, axis=-1, keepdims=True)
        n = nrm[idx]
        if mode1 == 0:
            col = n * 0.5 + 0.5
        else:
            vd = unit(cam - xyz[idx])
            if mode1 == 1:
                col = unit(vd - 2.0 * (n * vd).sum(-1, keepdims=True) * n) * 0.5 + 0.5
            elif mode1 == 2:
                w2c = np.linalg.inv(view.astype(np.float32))
                col = unit((cam + n) @ w2c[:3, :3].T + w2c[:3, 3]) * 0.5 + 0.5
            else:
                col = vd * 0.5 + 0.5
    return np.where(covered[..., None], col, 0).astype(np.float32)


def test_multiscale_render_tokens_of_the_dsl(hip):
    """MultiscaleRender over a mixed input format: ids with sizes, colours, normals variants, xyz, depth, labels, each at
    its own downscale, against the oracle rasteriser + the NumPy restatement of the vertex colours."""
    from read_amd.render import parse_input_string
    W, H, N = 192, 128, 120_000
    rng = np.random.default_rng(5)
    xyz = synthetic.make_cloud(N, seed=12)
    rgb = rng.random((N, 3)).astype(np.float32)
    nrm = rng.standard_normal((N, 3)).astype(np.float32)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    scene = Scene()
    scene.set_vertices(xyz, colors=rgb, normals=nrm, uv1d=np.arange(N))
    proj = synthetic.make_proj(W, H, f=120.0)
    view = synthetic.sweep_pose(8)
    fmt = ("uv_1d_p1, uv_1d_p3_ds1, uv_1d_ps30_ds2, colors_p1, colors_p2_ds1, normals_m_p1, normals_r_p1_ds1, "
           "normals_l_p2, normals_d_p1_ds2, xyz_p1_ds1, depth_p1, labels_p1_ds3")
    mr = MultiscaleRender(scene, fmt, (W, H), proj_matrix=proj, out_buffer_location='torch')
    out = mr.render(view_matrix=view)
    M = camera.total_matrix(proj, view)[0]
    for tok in fmt.replace(' ', '').split(','):
        cfg = parse_input_string(tok)
        s = cfg.get('downscale', 0)
        w, h = W >> s, H >> s
        oi, od = oracle.raster_level_gl(xyz, M, w, h, point_size=cfg['point_size'], relative=cfg['splat_mode'])
        got = out[tok].cpu().numpy()
        if cfg['mode'][0] == 3:
            ref = np.zeros((h, w, 3), np.float32)
            ref[..., 0] = oi.astype(np.float32)
            assert np.array_equal(got, ref), tok
        else:
            ref = _colour_oracle(scene, cfg['mode'], oi, od, M, view)
            if 'depth' in tok or 'label' in tok:
                ref = ref[..., :1]
            assert got.shape == ref.shape, tok
            np.testing.assert_allclose(got, ref, rtol=1e-5, atol=1e-6, err_msg=tok)
    # gl_frame flips rows (READ/datasets/dynamic.py:88-94), numpy buffers are host arrays
    mr2 = MultiscaleRender(scene, "colors_p1", (W, H), proj_matrix=proj, out_buffer_location='numpy', gl_frame=True)
    flipped = mr2.render(view_matrix=view)["colors_p1"]
    assert isinstance(flipped, np.ndarray) and np.array_equal(flipped[::-1], out["colors_p1"].cpu().numpy())
    # augmentation buffers apply to every token, including the id pyramid (which then leaves the single-pass path)
    mask = rng.random(N) < 0.4
    scene.set_point_discard(mask)
    scene.set_point_perturb_seeded(0.2, seed=3)
    fmt5 = "uv_1d_p1, uv_1d_p1_ds1, uv_1d_p1_ds2"
    out5 = MultiscaleRender(scene, fmt5, (W, H), proj_matrix=proj, out_buffer_location='torch').render(view_matrix=view)
    for l, tok in enumerate(fmt5.replace(' ', '').split(',')):
        oi, _ = oracle.raster_level_gl(xyz, M, W >> l, H >> l, discard=mask, perturb_hash=(0.2, 3))
        assert np.array_equal(out5[tok][..., 0].cpu().numpy(), oi.astype(np.float32)), tok
    with pytest.raises(NotImplementedError):
        MultiscaleRender(scene, "colors", (W, H), proj_matrix=proj).render(view_matrix=view)       # triangles
    with pytest.raises(NotImplementedError):
        MultiscaleRender(scene, "uv_2d_p1", (W, H), proj_matrix=proj).render(view_matrix=view)     # mesh texture coords


def test_ogl_supersampling_end_to_end(hip):
    """OGL(..., supersampling=2): raster at 2x, fused bilinear reduce of the descriptors, UNet at 1x — against the oracle
    (raster at 2x -> gather -> F.interpolate(1/2) -> UNet), both through the fast path and through NetAndTexture."""
    from read_amd.net_texture import NetAndTexture
    from read_amd.ogl import OGL
    from read_amd.texture import PointTexture
    from read_amd.unet import UNet
    from tests.unet_spec import UNET_SPEC
    W, H, N, ss = 96, 64, 60_000, 2
    fmt = "uv_1d_p1, uv_1d_p1_ds1, uv_1d_p1_ds2, uv_1d_p1_ds3, uv_1d_p1_ds4"
    xyz = synthetic.make_cloud(N, seed=31)
    state = synthetic.make_unet_state(UNET_SPEC, 8)
    net = UNet()
    net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in state.items()})
    tex = PointTexture(8, N, init_method='rand')
    model = NetAndTexture(net, {0: tex}, supersampling=ss)
    model.load_textures(0)
    scene = Scene(xyz)
    proj = synthetic.make_proj(W, H, f=70.0)
    scene.set_proj_matrix(proj)
    scene.set_camera_view(synthetic.sweep_pose(4))
    ogl = OGL.from_model(scene, model, fmt, (W, H), supersampling=ss)
    fast = ogl.infer()["output"]
    assert ogl.last_path == 'fast'                  # fused supersampling branch of the device-resident path
    M = camera.total_matrix(proj, synthetic.sweep_pose(4))[0]
    oi, _ = oracle.raster_multiscale(xyz, M, ss * W, ss * H, 5)
    desc = tex.texture_.detach().cpu().numpy()
    with torch.no_grad():
        feats = [F.interpolate(unet_torch.point_texture_forward(desc, i[None]), scale_factor=1. / ss, mode='bilinear')
                 for i in oi]
        ref = unet_torch.unet_forward(state, *feats[:4])[0]
    assert fast.shape == (H, W, 4)
    assert unet_torch.psnr(fast[..., :3].permute(2, 0, 1).cpu(), ref) >= 120.0
    # the dict path (MultiscaleRender at ss x -> NetAndTexture) gives the same frame
    inputs = {k: v.permute(2, 0, 1)[None] for k, v in ogl.renderer.render().items()}
    slow = ogl.infer(inputs)["output"]
    assert ogl.last_path == 'dict'
    torch.testing.assert_close(fast, slow, rtol=0, atol=1e-6)
