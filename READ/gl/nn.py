from read_amd.ogl import OGL  # noqa: F401
