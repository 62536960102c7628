"""Mint the scene-ingest fixtures and pin them with the REFERENCE's own loaders (run in the build container only).

    python tests/golden/make_scene_golden.py

Writes tests/golden/scene/{scene.yaml, scene_txt.yaml, camera.xml, pointcloud.ply, view_matrix.txt, proj_matrix.txt,
model3d_origin.txt} (synthetic, seeded) and tests/golden/scene_io.npz = what /root/reference/READ/gl/utils.py
returns for them.  The reference module imports cv2, trimesh and the GL scene at module level; none is installed
here, so they are stubbed — the functions pinned below never touch them, except ``import_model3d`` (trimesh.load),
which is replaced by read_amd.scene_io.import_model3d: the PLY decoding itself is pinned by
tests/test_scene_io.py against the PLY format (hand-written ASCII file, writer round trips).
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from read_amd import scene_io                                        # noqa: E402

D = os.path.join(HERE, "scene")
os.makedirs(D, exist_ok=True)
rng = np.random.default_rng(2019)

# ---- fixtures
n = 200
xyz = rng.uniform(-5, 5, (n, 3)).astype(np.float32)
nrm = rng.standard_normal((n, 3)).astype(np.float32)
nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
rgb = rng.integers(0, 256, (n, 3), dtype=np.uint8)
scene_io.write_ply(os.path.join(D, "pointcloud.ply"), xyz, rgb=rgb, normals=nrm)


def pose(k):
    a = 0.1 * k
    R = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
    m = np.eye(4)
    m[:3, :3] = R
    m[:3, 3] = [0.5 * k, -0.25 * k, 2.0 + k]
    return m


cams = []
for k, label in enumerate(["000010", "7", "cam_b", "unaligned"]):
    if label == "unaligned":
        cams.append(f'      <camera id="{k}" sensor_id="0" label="{label}"/>')
    else:
        cams.append(f'      <camera id="{k}" sensor_id="0" label="{label}">\n        <transform>'
                    + " ".join(repr(float(v)) for v in pose(k).reshape(-1)) + "</transform>\n      </camera>")
xml = f"""<?xml version="1.0" encoding="UTF-8"?>
<document version="1.4.0">
  <chunk label="Chunk 1" enabled="true">
    <sensors next_id="1">
      <sensor id="0" label="synthetic" type="frame">
        <resolution width="1216" height="368"/>
        <calibration type="frame" class="adjusted">
          <resolution width="1216" height="368"/>
          <f>718.856</f>
          <cx>-3.5</cx>
          <cy>1.25</cy>
        </calibration>
      </sensor>
    </sensors>
    <cameras next_id="4" next_group_id="0">
{chr(10).join(cams)}
    </cameras>
  </chunk>
</document>
"""
open(os.path.join(D, "camera.xml"), "w").write(xml)
vm = np.stack([pose(k) for k in range(4)])
vm[2, 1, 3] = np.nan                                                  # dropped by get_valid_matrices
np.savetxt(os.path.join(D, "view_matrix.txt"), vm.reshape(-1, 4))
pm = np.array([[1.18, 0, 0, 0], [0, 3.9, 0, 0], [0, 0, -1.0002, -0.2], [0, 0, -1, 0]])
np.savetxt(os.path.join(D, "proj_matrix.txt"), pm)
origin = np.eye(4)
origin[:3, 3] = [1, 2, 3]
np.savetxt(os.path.join(D, "model3d_origin.txt"), origin)
open(os.path.join(D, "scene.yaml"), "w").write(
    "viewport_size: [1216, 368]\nintrinsic_matrix: camera.xml\nview_matrix:  camera.xml\npointcloud: pointcloud.ply\n")
open(os.path.join(D, "scene_txt.yaml"), "w").write(
    "viewport_size: [640, 480]\nview_matrix: view_matrix.txt\nproj_matrix: proj_matrix.txt\n"
    "model3d_origin: model3d_origin.txt\npointcloud: pointcloud.ply\nnet_path: runs/x\nckpt: UNet_1.pth\n"
    "texture_ckpt: PointTexture_1.pth\n")

# ---- the reference's loaders
for name in ("cv2", "trimesh"):
    sys.modules.setdefault(name, types.ModuleType(name))
prog = types.ModuleType("READ.gl.programs")
prog.NNScene = object
sys.modules["READ.gl.programs"] = prog
import importlib.util                                                # noqa: E402
spec = importlib.util.spec_from_file_location("_reference_gl_utils", "/root/reference/READ/gl/utils.py")
ref = importlib.util.module_from_spec(spec)                          # by path: the repo's own READ/ alias package shadows
spec.loader.exec_module(ref)                                         # the reference's namespace package on sys.path
ref.import_model3d = scene_io.import_model3d

out = {}
K, wh = ref.intrinsics_from_xml(os.path.join(D, "camera.xml"))
out["K"], out["wh"] = K, np.array(wh)
vms, labels = ref.extrinsics_from_xml(os.path.join(D, "camera.xml"))
out["xml_view"], out["xml_labels"] = np.stack(vms), np.array(labels)
vms, labels = ref.extrinsics_from_view_matrix(os.path.join(D, "view_matrix.txt"))
out["txt_view"], out["txt_labels"] = np.stack(vms), np.array(labels)
out["recalc"] = ref.recalc_proj_matrix_planes(pm)
out["rescale"] = ref.rescale_K(K, 0.5, 0.25)
out["crop"] = ref.crop_intrinsic_matrix(K, (1216, 368), (512, 256))
out["xyz_c"] = ref.get_xyz_colors(xyz.astype(np.float64))
for tag, y in (("a", "scene.yaml"), ("b", "scene_txt.yaml")):
    sd = ref.load_scene_data(os.path.join(D, y))
    out[f"{tag}_keys"] = np.array(sorted(sd))
    out[f"{tag}_viewport"] = np.array(sd["config"]["viewport_size"])
    out[f"{tag}_view"] = np.stack(sd["view_matrix"])
    out[f"{tag}_labels"] = np.array(sd["camera_labels"])
    out[f"{tag}_origin"] = sd["model3d_origin"]
    out[f"{tag}_K"] = sd["intrinsic_matrix"] if sd["intrinsic_matrix"] is not None else np.zeros(0)
    out[f"{tag}_proj"] = sd["proj_matrix"] if sd["proj_matrix"] is not None else np.zeros(0)
    out[f"{tag}_ckpt"] = np.array([str(sd["net_ckpt"]), str(sd["tex_ckpt"])])
np.savez_compressed(os.path.join(HERE, "scene_io.npz"), **out)
print("wrote", os.path.join(HERE, "scene_io.npz"), {k: v.shape for k, v in out.items()})
