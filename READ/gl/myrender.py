from read_amd.render import MyRender  # noqa: F401
