"""Alias of the reference module path READ/gl/utils.py for the functions on the render path and the scene ingest."""
from read_amd.camera import get_proj_matrix  # noqa: F401
from read_amd.scene_io import (crop_intrinsic_matrix, extrinsics_from_view_matrix, extrinsics_from_xml,  # noqa: F401
                               fix_relative_path, get_valid_matrices, get_xyz_colors, import_model3d,
                               intrinsics_from_xml, load_scene, load_scene_data, recalc_proj_matrix_planes,
                               rescale_K, setup_scene)
