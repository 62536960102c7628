"""``READ.datasets.dynamic`` with the GL-free ``MultiscaleRender`` (READ/datasets/dynamic.py:50-99).

When the reference checkout is on ``sys.path`` behind this repo, its own ``READ/datasets/dynamic.py`` is executed first
(so ``get_datasets``, ``DynamicDataset`` ... stay available to ``train.py`` and ``TexturePipeline.create``) and the names its
``DynamicDataset`` resolves through ITS module globals at run time are re-bound there: ``MultiscaleRender`` (dynamic.py:197)
and ``NNScene`` (:173) to the HIP-backed classes, ``app.Window`` (:196, "creates GL context") to a no-op.  Its module-level
imports that only exist for OpenGL (``glumpy``, ``READ.gl.render``) are satisfied by inert stand-ins when the real packages
are absent; ``READ.gl.programs`` / ``READ.gl.utils`` / ``READ.gl.dataset`` resolve to this alias package.  If the reference
module still cannot be executed (no checkout, or torchvision / cv2 missing) this module exports the renderer alone and
``reference_origin`` says why."""
import importlib.util
import os
import sys
import types

from read_amd.render import MultiscaleRender as _HipMultiscaleRender
from read_amd.render import Scene as _HipScene


class _NoGLWindow:
    """``app.Window(visible=False)`` creates the GL context in the reference; nothing to create here."""

    def __init__(self, *args, **kwargs):
        pass


def _inert_gl_modules():
    """glumpy / READ.gl.render stand-ins, installed only when the real ones cannot be imported."""
    added = []
    try:
        import glumpy  # noqa: F401
    except Exception:
        g = types.ModuleType("glumpy")
        g.app = types.SimpleNamespace(Window=_NoGLWindow)
        g.gl = types.SimpleNamespace()
        sys.modules["glumpy"] = g
        added.append("glumpy")
    if "READ.gl.render" not in sys.modules:
        try:
            importlib.import_module("READ.gl.render")
        except Exception:
            r = types.ModuleType("READ.gl.render")

            class OffscreenRender:                         # the GL framebuffer renderer is replaced, never constructed
                def __init__(self, *a, **k):
                    raise RuntimeError("READ.gl.render.OffscreenRender needs OpenGL; the render path uses read_amd instead")
            r.OffscreenRender = OffscreenRender
            sys.modules["READ.gl.render"] = r
            added.append("READ.gl.render")
    return added


def _load_reference_module():
    here = os.path.dirname(os.path.abspath(__file__))
    for entry in sys.path:
        cand = os.path.join(entry or ".", "READ", "datasets", "dynamic.py")
        if os.path.isfile(cand) and os.path.dirname(os.path.abspath(cand)) != here:
            added = _inert_gl_modules()
            spec = importlib.util.spec_from_file_location(__name__ + "._reference", cand)
            mod = importlib.util.module_from_spec(spec)
            try:
                spec.loader.exec_module(mod)
            except Exception as e:                         # e.g. torchvision / cv2 not installed
                for name in added:
                    sys.modules.pop(name, None)
                return None, f"{cand}: {type(e).__name__}: {e}"
            return mod, cand
    return None, "no other READ/datasets/dynamic.py on sys.path"


_ref, reference_origin = _load_reference_module()
if _ref is not None:
    # DynamicDataset.load / __getitem__ look these up in the reference module's own globals
    _ref.MultiscaleRender = _HipMultiscaleRender
    _ref.NNScene = _HipScene
    _ref.app = types.SimpleNamespace(Window=_NoGLWindow)
    globals().update({k: v for k, v in vars(_ref).items() if not k.startswith("__")})
MultiscaleRender = _HipMultiscaleRender
