"""CPU: the torch-fp32 UNet restatement against the golden frame produced by the REFERENCE's own
modules (tests/golden/make_golden.py).  Same torch build -> agreement to a few ULP; the tolerance
leaves room for a different CPU's oneDNN kernel choice."""
import os

import numpy as np
import torch

import oracle
from oracle import unet_torch
from read_amd import camera, synthetic
from tests.unet_spec import UNET_SPEC


def test_spec_shape():
    assert len(UNET_SPEC) == 101
    n_params = sum(2 * (co * ci * k * k + co) + 2 * co for (_, ci, co, k) in UNET_SPEC)
    assert n_params == 30_193_988


def test_golden_frame(golden_dir):
    g = np.load(os.path.join(golden_dir, "frame_64x48.npz"))
    W, H, N, seed = int(g["W"]), int(g["H"]), int(g["N"]), int(g["seed"])
    xyz, desc = synthetic.make_cloud(N, seed), synthetic.make_descriptors(N, 8, seed)
    state = synthetic.make_unet_state(UNET_SPEC, seed)
    M = camera.total_matrix(synthetic.make_proj(W, H, f=float(g["f"])), synthetic.sweep_pose(int(g["pose"])))[0]
    assert np.array_equal(M, g["M"])
    idx, dep = oracle.raster_multiscale(xyz, M, W, H, 5)
    for l in range(5):
        assert np.array_equal(idx[l], g[f"idx{l}"])
        assert np.array_equal(dep[l].view(np.uint32), g[f"depth{l}"].view(np.uint32))
    taps = {}
    with torch.no_grad():
        feats = [unet_torch.point_texture_forward(desc[None], i[None]) for i in idx]
        rgb = unet_torch.unet_forward(state, *feats[:4], taps=taps)[0]
    torch.testing.assert_close(rgb, torch.from_numpy(g["rgb"]), rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(taps["zb"][0], torch.from_numpy(g["zb"]), rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(taps["res1"][0][:, ::4, ::4], torch.from_numpy(g["res1"]), rtol=1e-4, atol=1e-4)


def test_psnr_definition():
    a = torch.zeros(3, 4, 4)
    b = torch.full((3, 4, 4), 0.1)
    assert abs(unet_torch.psnr(a, b) - 20.0) < 1e-4
    assert unet_torch.psnr(a, a) == float("inf")


def test_training_mode_equals_reference_module_in_train():
    """Live pin (only where /root/reference exists): the oracle with training=True against the reference's own UNet in
    .train() — batch-statistics BatchNorm output, its gradients and the running-buffer update (READ/models/unet.py:40,51;
    train.py:271-279 puts the model in .train() unless eval_in_train)."""
    import sys
    import types
    import pytest
    ref = "/root/reference"
    if not os.path.isdir(os.path.join(ref, "READ", "models")):
        pytest.skip("reference checkout not present")
    import importlib.util
    spec = importlib.util.spec_from_file_location("_ref_unet_for_pin", os.path.join(ref, "READ", "models", "unet.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    state = synthetic.make_unet_state(UNET_SPEC, 5)
    net = mod.UNet()
    net.load_state_dict({k: torch.from_numpy(np.asarray(v)).clone() for k, v in state.items()}, strict=True)
    net.train()
    rng = np.random.default_rng(8)
    xs = [torch.from_numpy(rng.random((2, 8, 32 >> l, 48 >> l)).astype(np.float32)) for l in range(4)]
    out_ref = net(*xs)
    g = torch.from_numpy(rng.standard_normal(tuple(out_ref.shape)).astype(np.float32))
    out_ref.backward(g)
    st = {k: (torch.from_numpy(np.asarray(v)).clone().requires_grad_(True)
              if (np.asarray(v).dtype == np.float32 and "running" not in k) else torch.from_numpy(np.asarray(v)).clone())
          for k, v in state.items()}
    out = unet_torch.unet_forward(st, *xs, training=True)
    out.backward(g)
    torch.testing.assert_close(out, out_ref, rtol=1e-5, atol=1e-5)
    sd = net.state_dict()
    for k in ("feat_extract.0.block.norm.running_mean", "Encoder.2.layers.1.main.0.block.norm.running_var",
              "SCM0.conv.block.norm.running_var", "feat_extract.5.block.norm.running_mean"):
        torch.testing.assert_close(st[k], sd[k], rtol=1e-5, atol=1e-6)
        assert not torch.equal(st[k], torch.from_numpy(np.asarray(state[k])))          # the buffers did move
    for name, p in net.named_parameters():
        if name.startswith("ConvsOut."):
            continue
        torch.testing.assert_close(st[name].grad, p.grad, rtol=1e-3, atol=1e-5 * float(p.grad.abs().max()) + 1e-12)
    # eval mode is untouched by the flag's plumbing
    with torch.no_grad():
        a = unet_torch.unet_forward(state, *xs)
        b = unet_torch.unet_forward(state, *xs, training=False)
    assert torch.equal(a, b)


def test_per_item_training_loop_equals_reference_net_and_texture_in_train():
    """the reference's NetAndTexture.forward calls the net once per batch item (READ/models/compose.py:
    137-176), so in .train() nn.BatchNorm2d sees N = 1: per-item statistics, running buffers moved B times per step,
    num_batches_tracked + B.  Live pin of ``oracle.unet_torch.net_and_texture_forward_batch(training=True)`` against the
    reference's own NetAndTexture + PointTexture + UNet in .train(): output, descriptor / parameter gradients, buffers — and
    that the result DIFFERS from one joint batch (what the round-3 oracle pinned)."""
    import sys
    import types
    import pytest
    ref = "/root/reference"
    if not os.path.isdir(os.path.join(ref, "READ", "models")):
        pytest.skip("reference checkout not present")
    import importlib.util

    def load(rel, name):
        added = [m for m in ("imageio", "cv2") if m not in sys.modules]
        for m in added:
            sys.modules[m] = types.ModuleType(m)
        try:
            spec = importlib.util.spec_from_file_location(name, os.path.join(ref, rel))
            mod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mod)
        finally:
            for m in added:
                sys.modules.pop(m, None)
        return mod
    r_unet, r_tex, r_comp = (load("READ/models/unet.py", "_ref_unet_pi"), load("READ/models/texture.py", "_ref_tex_pi"),
                             load("READ/models/compose.py", "_ref_comp_pi"))
    state = synthetic.make_unet_state(UNET_SPEC, 9)
    N, B, H, W = 500, 3, 32, 48
    rng = np.random.default_rng(10)
    desc = rng.random((1, 8, N)).astype(np.float32)
    net = r_unet.UNet()
    net.load_state_dict({k: torch.from_numpy(np.asarray(v)).clone() for k, v in state.items()}, strict=True)
    tex = r_tex.PointTexture(8, N, activation='none')
    with torch.no_grad():
        tex.texture_.copy_(torch.from_numpy(desc))
    model = r_comp.NetAndTexture(net, {0: tex})
    model.load_textures(0)
    model.train()
    keys = ["uv_1d_p1", "uv_1d_p1_ds1", "uv_1d_p1_ds2", "uv_1d_p1_ds3", "uv_1d_p1_ds4"]
    maps = [rng.integers(0, N, (B, H >> l, W >> l)) for l in range(5)]
    inputs = {'id': torch.zeros(B, dtype=torch.long)}
    inputs.update({k: torch.from_numpy(m[:, None].astype(np.float32)) for k, m in zip(keys, maps)})
    out_ref = model(inputs)
    g = torch.from_numpy(rng.standard_normal(tuple(out_ref.shape)).astype(np.float32))
    out_ref.backward(g)

    def fresh():
        return {k: (torch.from_numpy(np.asarray(v)).clone().requires_grad_(True)
                    if (np.asarray(v).dtype == np.float32 and "running" not in k) else torch.from_numpy(np.asarray(v)).clone())
                for k, v in state.items()}
    st = fresh()
    d = torch.from_numpy(desc.copy()).requires_grad_(True)
    out = unet_torch.net_and_texture_forward_batch(st, d, maps, training=True)
    out.backward(g)
    torch.testing.assert_close(out, out_ref, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(d.grad, tex.texture_.grad, rtol=1e-4, atol=1e-6 * float(tex.texture_.grad.abs().max()))
    sd = net.state_dict()
    for k in sd:
        if "running_" in k and not k.startswith("ConvsOut."):
            torch.testing.assert_close(st[k], sd[k], rtol=1e-5, atol=1e-6)
        if k.endswith("num_batches_tracked") and not k.startswith("ConvsOut."):
            assert int(sd[k]) == B                              # one BatchNorm call per item
    for name, p in net.named_parameters():
        if not name.startswith("ConvsOut."):
            torch.testing.assert_close(st[name].grad, p.grad, rtol=1e-3, atol=1e-5 * float(p.grad.abs().max()) + 1e-12)
    # one joint batch is a different function: statistics over all B items, buffers moved once
    st_j = fresh()
    feats = [unet_torch.point_texture_forward(desc, m) for m in maps[:4]]
    out_j = unet_torch.unet_forward(st_j, *feats, training=True)
    assert float((out_j - out_ref).abs().max()) > 1e-3
    k = "feat_extract.0.block.norm.running_mean"
    assert float((st_j[k] - sd[k]).abs().max()) > 1e-6
    # ... and the feature-pyramid form of the loop is the same function
    st2 = fresh()
    out2 = unet_torch.unet_forward_per_item(st2, *feats, training=True)
    assert torch.equal(out2, out.detach()) or float((out2 - out).abs().max()) < 1e-6
