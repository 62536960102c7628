"""ORACLE — TEST INFRASTRUCTURE ONLY.

CPU restatements of READ's per-frame render path, used as the parity checker by
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg.
Nothing under ``read_amd/`` may import this package: the product path is the HIP
library and must fail loudly without it.

Contents
--------
raster.c        C restatement of the z-buffer projector (point_render.cu:96-200)
raster_np.py    independent NumPy restatement of the same (cross-check of raster.c)
unet_torch.py   functional torch-fp32 restatement of READ/models/unet.py + texture.py + compose.py
build.py        compiles raster.c -> oracle/liboracle_raster.so (gcc, -ffp-contract=off)
build_ref.sh    compiles the reference's own DepthProject source for the CPU -> oracle/_ref/
"""
from .raster_c import (  # noqa: F401
    raster_level, raster_multiscale, index_to_float, gather_chw, gather_backward_chw, lib_path,
    raster_level_gl, drop_mask, perturb_array, drop_threshold, project_points,
)
