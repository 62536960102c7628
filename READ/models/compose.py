from read_amd.net_texture import NetAndTexture  # noqa: F401
