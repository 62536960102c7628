"""Alias of READ/models/texture.py: ``PointTexture`` / ``Texture`` are the HIP-backed ones; ``MeshTexture`` (:73-159, the
mip-mapped 2-D texture of the mesh pipelines) is outside the point-cloud render path and refuses to be built."""
from read_amd.texture import PointTexture, Texture  # noqa: F401


class MeshTexture(Texture):
    def __init__(self, *args, **kwargs):
        raise NotImplementedError("MeshTexture (use_mesh / RGBTexturePipeline) draws triangles: outside the point-cloud "
                                  "render path this repo rebuilds (DESIGN.md section 6)")
