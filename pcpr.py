"""Top-level alias so that the reference's ``import pcpr`` (src/READ/gl/myrender.py:3) binds the
MI355X rasteriser: ``pcpr.forward(points, total_m, w, h, block)``."""
from read_amd.pcpr import clear_cache, forward  # noqa: F401
