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
