"""``READ.datasets.dynamic`` with the GL-free ``MultiscaleRender`` (READ/datasets/dynamic.py:50-99).

When the reference checkout is on ``sys.path`` behind this repo, its own ``READ/datasets/dynamic.py`` is executed into this
module first (so ``get_datasets``, ``DynamicDataset`` ... stay available to ``train.py``) and only the renderer class is
replaced; if that module cannot be imported (no checkout, or its OpenGL dependencies are missing) this module exports the
renderer alone."""
import importlib.util
import os
import sys

from read_amd.render import MultiscaleRender as _HipMultiscaleRender


def _load_reference_module():
    here = os.path.dirname(os.path.abspath(__file__))
    for entry in sys.path:
        cand = os.path.join(entry or ".", "READ", "datasets", "dynamic.py")
        if os.path.isfile(cand) and os.path.dirname(os.path.abspath(cand)) != here:
            spec = importlib.util.spec_from_file_location(__name__ + "._reference", cand)
            mod = importlib.util.module_from_spec(spec)
            try:
                spec.loader.exec_module(mod)
            except Exception as e:                         # e.g. glumpy / OpenGL not installed
                return None, f"{cand}: {type(e).__name__}: {e}"
            return mod, cand
    return None, "no other READ/datasets/dynamic.py on sys.path"


_ref, reference_origin = _load_reference_module()
if _ref is not None:
    globals().update({k: v for k, v in vars(_ref).items() if not k.startswith("__")})
MultiscaleRender = _HipMultiscaleRender
