"""Generate the committed golden vectors by running the REFERENCE's own Python modules
(imported from /root/reference; this only works in the build container) on seeded inputs.

    python tests/golden/make_golden.py

Writes tests/golden/*.npz.  Also asserts that the oracle restatement (oracle/unet_torch.py)
reproduces the reference on the same inputs, and that the oracle rasteriser reproduces the
reference's DepthProject source executed serially (oracle/_ref).
Weights are NOT stored (30 M parameters): they come from read_amd.synthetic.make_unet_state,
a NumPy-seeded recipe that yields the same bytes on any machine.
"""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))

from read_amd import synthetic, camera          # noqa: E402
import oracle                                   # noqa: E402
from oracle import unet_torch, ref_c            # noqa: E402

# layer spec without needing the HIP library: (path, cin, cout, k) — must equal read_amd.unet.weight_spec()
from tests.unet_spec import UNET_SPEC           # noqa: E402


def import_reference():
    for stub in ("imageio", "cv2"):             # imported but unused by READ/models/compose.py:6-7
        sys.modules.setdefault(stub, types.ModuleType(stub))
    sys.path.insert(0, REF)
    from READ.models.unet import UNet
    from READ.models.texture import PointTexture
    from READ.models.compose import NetAndTexture
    return UNet, PointTexture, NetAndTexture


def main():
    torch.set_num_threads(os.cpu_count())
    UNet, PointTexture, NetAndTexture = import_reference()
    seed = synthetic.DEFAULT_SEED

    # ---------------- UNet + texture + compose golden (64x48 frame, 5000 points)
    W, H, N = 64, 48, 5000
    state = synthetic.make_unet_state(UNET_SPEC, seed)
    net = UNet()
    missing = net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in state.items()}, strict=True)
    net.eval()
    xyz = synthetic.make_cloud(N, seed)
    desc = synthetic.make_descriptors(N, 8, seed)                      # (C,N)
    proj = synthetic.make_proj(W, H, f=40.0)
    M = camera.total_matrix(proj, synthetic.sweep_pose(3))[0]
    idx, dep = oracle.raster_multiscale(xyz, M, W, H, 5)
    tex = PointTexture(8, N, activation='none', init_method='zeros')
    with torch.no_grad():
        tex.texture_.copy_(torch.from_numpy(desc)[None])
    model = NetAndTexture(net, {0: tex})
    model.load_textures(0)
    model.eval()
    inputs = {'id': 0}
    for l in range(5):
        key = 'uv_1d_p1' if l == 0 else f'uv_1d_p1_ds{l}'
        f = torch.from_numpy(oracle.index_to_float(idx[l]))[None, None]
        inputs[key] = f
    with torch.no_grad():
        feats_ref = [tex(inputs[k]) for k in list(inputs)[1:]]
        taps_ref = {}
        out_ref = model(dict(inputs))                                   # reference end-to-end (compose.py)
    # oracle restatement must reproduce the reference
    with torch.no_grad():
        taps = {}
        feats_or = [unet_torch.point_texture_forward(desc[None], idx[l][None]) for l in range(5)]
        out_or = unet_torch.unet_forward(state, *feats_or[:4], taps=taps)
    for a, b in zip(feats_ref, feats_or):
        assert torch.equal(a, b), "gather restatement differs from reference PointTexture"
    err = (out_ref - out_or).abs().max().item()
    print(f"UNet oracle vs reference: max|diff| = {err:.3e}, PSNR = {unet_torch.psnr(out_ref, out_or):.1f} dB")
    assert err < 1e-4, err
    np.savez_compressed(
        os.path.join(HERE, "frame_64x48.npz"), W=W, H=H, N=N, seed=seed, f=40.0, pose=3, M=M,
        **{f"idx{l}": idx[l] for l in range(5)}, **{f"depth{l}": dep[l] for l in range(5)},
        rgb=out_ref[0].numpy(),
        res1=taps["res1"][0].numpy()[:, ::4, ::4], zb=taps["zb"][0].numpy(), z8=taps["z8"][0].numpy(),
        aff2=taps["aff2"][0].numpy()[:, ::2, ::2])

    # ---------------- rasteriser golden: reference DepthProject source run serially on the CPU
    W, H, N = 256, 256, 100000                                          # BASELINE.json configs[0]
    xyz = synthetic.make_cloud(N, seed)
    proj = synthetic.make_proj(W, H, f=256.0)
    Ms = camera.total_matrix(proj, np.stack([np.eye(4, dtype=np.float32), synthetic.sweep_pose(40)]))
    gold = {}
    for l, (w, h) in enumerate(camera.level_sizes(W, H, 5)):
        ri, rd = ref_c.pcpr_ref_forward(xyz, Ms, w, h)
        for b in range(2):
            oi, od = oracle.raster_level(xyz, Ms[b], w, h)
            assert np.array_equal(oracle.index_to_float(oi), ri[b]) and np.array_equal(od.view(np.uint32), rd[b].view(np.uint32))
        gold[f"index{l}"] = ri.astype(np.int32)
        gold[f"depth{l}"] = rd
    np.savez_compressed(os.path.join(HERE, "raster_256_100k.npz"), W=W, H=H, N=N, seed=seed, f=256.0, M=Ms, **gold)

    # ---------------- get_proj_matrix golden: exec the reference function text (utils.py imports cv2/trimesh)
    src = open(os.path.join(REF, "READ/gl/utils.py")).read().splitlines()[122:150]
    ns = {"np": np}
    exec("\n".join(src), ns)
    K = synthetic.make_intrinsics(1216, 352)
    Pref = ns["get_proj_matrix"](K, (1216, 352), 0.1, 1000.0)
    assert np.array_equal(Pref, camera.get_proj_matrix(K, (1216, 352), 0.1, 1000.0))
    np.savez_compressed(os.path.join(HERE, "proj_1216x352.npz"), K=K, P=Pref, znear=0.1, zfar=1000.0)
    print("golden vectors written to", HERE)


if __name__ == "__main__":
    main()
