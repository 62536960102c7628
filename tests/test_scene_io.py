"""CPU: scene / camera ingest — the native loaders against what the reference's own
READ/gl/utils.py returned for the committed fixtures (tests/golden/make_scene_golden.py), and the PLY reader
against the PLY format (hand-written ASCII file, all three encodings through the writer)."""
import os

import numpy as np
import pytest

from read_amd import scene_io

SCENE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "scene")


@pytest.fixture(scope="module")
def golden(golden_dir):
    return np.load(os.path.join(golden_dir, "scene_io.npz"))


def test_camera_xml_matches_reference(golden):
    K, wh = scene_io.intrinsics_from_xml(os.path.join(SCENE, "camera.xml"))
    assert K.dtype == np.float32 and np.array_equal(K, golden["K"]) and tuple(wh) == tuple(golden["wh"])
    assert K[0, 2] == 608.0 and K[1, 2] == 184.0            # principal point = image centre, cx/cy of the file ignored
    vms, labels = scene_io.extrinsics_from_xml(os.path.join(SCENE, "camera.xml"))
    assert labels == list(golden["xml_labels"]) == ["000010", "7", "cam_b"]       # file order, unaligned camera dropped
    assert np.array_equal(np.stack(vms), golden["xml_view"])
    assert vms[1][1, 1] == -1.0 and vms[1][2, 3] == 3.0      # columns 1, 2 negated (GL axes), translation kept


def test_view_matrix_text_file_matches_reference(golden):
    vms, labels = scene_io.extrinsics_from_view_matrix(os.path.join(SCENE, "view_matrix.txt"))
    assert labels == list(golden["txt_labels"]) == ["0", "1", "3"]                # the NaN matrix is dropped
    assert np.array_equal(np.stack(vms), golden["txt_view"])


def test_matrix_helpers_match_reference(golden):
    pm = np.loadtxt(os.path.join(SCENE, "proj_matrix.txt"))
    assert np.array_equal(scene_io.recalc_proj_matrix_planes(pm), golden["recalc"])
    assert np.array_equal(scene_io.rescale_K(golden["K"], 0.5, 0.25), golden["rescale"])
    assert np.array_equal(scene_io.crop_intrinsic_matrix(golden["K"], (1216, 368), (512, 256)), golden["crop"])
    xyz = scene_io.import_model3d(os.path.join(SCENE, "pointcloud.ply"))["xyz"]
    assert np.array_equal(scene_io.get_xyz_colors(xyz), golden["xyz_c"])


@pytest.mark.parametrize("tag,yaml_name", [("a", "scene.yaml"), ("b", "scene_txt.yaml")])
def test_load_scene_data_matches_reference(golden, tag, yaml_name):
    sd = scene_io.load_scene_data(os.path.join(SCENE, yaml_name))
    assert sorted(sd) == list(golden[f"{tag}_keys"])
    assert sd["config"]["viewport_size"] == tuple(golden[f"{tag}_viewport"])
    assert np.array_equal(np.stack(sd["view_matrix"]), golden[f"{tag}_view"])
    assert list(sd["camera_labels"]) == list(golden[f"{tag}_labels"])
    assert np.array_equal(sd["model3d_origin"], golden[f"{tag}_origin"])
    for key, g in (("intrinsic_matrix", f"{tag}_K"), ("proj_matrix", f"{tag}_proj")):
        if golden[g].size:
            assert np.array_equal(sd[key], golden[g])
        else:
            assert sd[key] is None
    assert [str(sd["net_ckpt"]), str(sd["tex_ckpt"])] == list(golden[f"{tag}_ckpt"])
    pc = sd["pointcloud"]
    assert pc["xyz"].shape == (200, 3) and pc["uv1d"].tolist() == list(range(200))
    assert pc["rgb"].min() >= 0 and pc["rgb"].max() <= 1 and pc["normals"].dtype == np.float32
    assert pc["uv2d"].shape == (200, 2) and pc["faces"].tolist() == [0, 1, 2]


def test_ply_ascii_by_hand(tmp_path):
    p = tmp_path / "tiny.ply"
    p.write_text("ply\nformat ascii 1.0\ncomment two points and a face\nelement vertex 2\nproperty double x\n"
                 "property double y\nproperty double z\nproperty uchar red\nproperty uchar green\nproperty uchar blue\n"
                 "element face 1\nproperty list uchar int vertex_indices\nend_header\n"
                 "0.5 -1.25 3 255 0 51\n1e-3 2 -4.5 0 128 255\n3 0 1 1\n")
    ply = scene_io.read_ply(str(p))
    assert ply["vertex"]["x"].dtype == np.float64 and ply["vertex"]["x"].tolist() == [0.5, 1e-3]
    assert ply["vertex"]["blue"].dtype == np.uint8 and ply["vertex"]["blue"].tolist() == [51, 255]
    assert ply["face"]["vertex_indices"].tolist() == [[0, 1, 1]]
    m = scene_io.import_model3d(str(p))
    assert m["xyz"].tolist() == [[0.5, -1.25, 3.0], [1e-3, 2.0, -4.5]]
    assert m["rgb"].tolist() == [[0.0, 0.0, 0.0], [0.0, 0.0, 0.0]]          # no normals -> the reference zeroes rgb (utils.py:456-458)


@pytest.mark.parametrize("fmt", ["ascii", "binary_little_endian", "binary_big_endian"])
def test_ply_round_trip_all_encodings(tmp_path, fmt):
    rng = np.random.default_rng(3)
    xyz = rng.standard_normal((257, 3)).astype(np.float32)
    nrm = rng.standard_normal((257, 3)).astype(np.float32)
    rgb = rng.integers(0, 256, (257, 3), dtype=np.uint8)
    p = str(tmp_path / f"c_{fmt}.ply")
    scene_io.write_ply(p, xyz, rgb=rgb, normals=nrm, fmt=fmt)
    m = scene_io.import_model3d(p)
    assert np.array_equal(m["xyz"].astype(np.float32), xyz)                  # float32 survives every encoding exactly
    assert np.array_equal(m["normals"], nrm)
    assert np.allclose(m["rgb"] * 255.0, rgb)
    scene_io.write_ply(p, xyz, fmt=fmt)                                      # positions only
    m = scene_io.import_model3d(p)
    assert np.array_equal(m["xyz"].astype(np.float32), xyz) and not m["rgb"].any()


def test_ply_errors(tmp_path):
    bad = tmp_path / "bad.ply"
    bad.write_text("plx\n")
    with pytest.raises(scene_io.PlyError):
        scene_io.read_ply(str(bad))
    bad.write_text("ply\nformat binary_little_endian 1.0\nelement vertex 4\nproperty float x\nproperty float y\n"
                   "property float z\nend_header\nabc")
    with pytest.raises(scene_io.PlyError):
        scene_io.read_ply(str(bad))
    bad.write_text("ply\nformat ascii 1.0\nelement vertex 1\nproperty float x\nproperty float y\nend_header\n1 2\n")
    with pytest.raises(scene_io.PlyError):
        scene_io.import_model3d(str(bad))
    with pytest.raises(NotImplementedError):
        scene_io.import_model3d(str(bad), is_mesh=True)


def test_setup_scene_feeds_the_render_scene():
    from read_amd.render import Scene
    sd = scene_io.load_scene_data(os.path.join(SCENE, "scene_txt.yaml"))
    scene = Scene()
    scene_io.setup_scene(scene, sd)
    assert scene.xyz.shape == (200, 3) and scene.xyz.dtype == np.float32
    assert np.array_equal(scene.view_matrix, np.asarray(sd["view_matrix"][0], np.float32))
    assert np.array_equal(scene.proj_matrix, sd["proj_matrix"].astype(np.float32))
    assert scene.model_matrix[:3, 3].tolist() == [1.0, 2.0, 3.0]
    with pytest.raises(NotImplementedError):
        scene_io.setup_scene(scene, sd, use_mesh=True)
