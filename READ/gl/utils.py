"""Alias of the reference module path READ/gl/utils.py for the functions on the render path and the scene ingest.
Names this repo does not rebuild (``pca_color``, ``nearest_train``, ``cv2_write`` ... — viewer / tooling helpers that need
cv2, trimesh or sklearn) are looked up lazily in the reference's own ``READ/gl/utils.py`` when that checkout sits behind
this repo on ``sys.path``."""
import importlib.util
import os
import sys

from read_amd.camera import get_proj_matrix  # noqa: F401
from read_amd.scene_io import (FastRand, crop_intrinsic_matrix, crop_proj_matrix, extrinsics_from_view_matrix,  # noqa: F401
                               extrinsics_from_xml, fix_relative_path, get_normal_colors, get_valid_matrices,
                               get_xyz_colors, import_model3d, intrinsics_from_xml, load_scene, load_scene_data,
                               recalc_proj_matrix_planes, rescale_K, setup_scene)

_reference = None


def _reference_module():
    global _reference
    if _reference is None:
        here = os.path.dirname(os.path.abspath(__file__))
        for entry in sys.path:
            cand = os.path.join(entry or ".", "READ", "gl", "utils.py")
            if os.path.isfile(cand) and os.path.dirname(os.path.abspath(cand)) != here:
                spec = importlib.util.spec_from_file_location(__name__ + "._reference", cand)
                mod = importlib.util.module_from_spec(spec)
                spec.loader.exec_module(mod)               # ImportError here names the missing third-party package
                _reference = mod
                break
        else:
            _reference = False
    return _reference


def __getattr__(name):
    if name.startswith("__"):
        raise AttributeError(name)
    ref = _reference_module()
    if ref and hasattr(ref, name):
        return getattr(ref, name)
    raise AttributeError(f"READ.gl.utils.{name} is not part of the MI355X render path and no reference checkout behind this "
                         f"repo on sys.path provides it")
