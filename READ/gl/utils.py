"""Alias of the reference module path READ/gl/utils.py for the functions on the render path and the scene ingest.
Names this repo does not rebuild (``pca_color``, ``nearest_train``, ``cv2_write`` ... — viewer / tooling helpers that need
cv2, trimesh or sklearn) are looked up lazily in the reference's own ``READ/gl/utils.py`` when that checkout sits behind
this repo on ``sys.path``."""
from read_amd._alias import lazy_reference_getattr
from read_amd.camera import get_proj_matrix  # noqa: F401
from read_amd.scene_io import (FastRand, crop_intrinsic_matrix, crop_proj_matrix, extrinsics_from_view_matrix,  # noqa: F401
                               extrinsics_from_xml, fix_relative_path, get_normal_colors, get_valid_matrices,
                               get_xyz_colors, import_model3d, intrinsics_from_xml, load_scene, load_scene_data,
                               recalc_proj_matrix_planes, rescale_K, setup_scene)

__getattr__ = lazy_reference_getattr(__name__, "READ/gl/utils.py")   # an ImportError there names the missing package
