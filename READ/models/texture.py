from read_amd.texture import PointTexture, Texture  # noqa: F401
