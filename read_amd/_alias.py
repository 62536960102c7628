"""Helpers of the ``READ/`` alias package (INTEGRATION.md level 1): find and execute the reference's module of the same
dotted name from the checkout that sits BEHIND this repo on ``sys.path``, so that an alias module can re-export every name the
MI355X path does not replace (``READ.models.compose.BoxFilter``, ``READ.pipelines.ogl.Pix2PixPipeline``, ``READ.gl.utils.
pca_color`` ...) instead of shadowing them away, and tell which of the reference's two trees that checkout is.

The reference keeps two variants of its package: the root tree (``READ/``: the net returns a tensor, ``viewer.py`` lives there)
and the ``src`` tree (``src/READ/``: the net returns ``{'im_out': tensor}``, ``src/READ/models/unet.py:280``; ``ModelAndLoss``
returns a dict of losses, ``src/READ/models/compose.py:29-40``; ``MyRender`` + the headless ``src/train.py`` exist only there).
``result_convention()`` says which one is behind: ``'tensor'`` or ``'dict'``."""
import importlib.util
import os
import sys
import types

_cache = {}
_convention = None


def _own_root():
    return os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def reference_file(relpath):
    """First ``<entry>/<relpath>`` on sys.path that is not this repo's own file, or None."""
    own = os.path.join(_own_root(), relpath)
    for entry in sys.path:
        cand = os.path.abspath(os.path.join(entry or ".", relpath))
        if os.path.isfile(cand) and cand != own:
            return cand
    return None


def _inert(name):
    """A stand-in for a third-party package the reference imports at module top and never needs on this path (``cv2``,
    ``imageio`` in compose.py:6-7): attribute access works, calling anything raises with the package's name."""
    m = types.ModuleType(name)

    def __getattr__(attr):
        if attr.startswith("__"):
            raise AttributeError(attr)

        def missing(*a, **k):
            raise ImportError(f"{name}.{attr} was called, but the package '{name}' is not installed")
        return missing
    m.__getattr__ = __getattr__
    return m


def reference_module(alias_name, relpath, optional_packages=()):
    """Execute the reference's ``relpath`` (e.g. ``READ/models/compose.py``) from behind this repo as
    ``<alias_name>._reference`` and return ``(module | None, origin-or-reason)``.  ``optional_packages`` that cannot be imported
    are replaced by inert stand-ins for the duration of the import only."""
    if relpath in _cache:
        return _cache[relpath]
    path = reference_file(relpath)
    if path is None:
        res = (None, f"no other {relpath} on sys.path")
    else:
        added = []
        for pkg in optional_packages:
            if pkg in sys.modules:
                continue
            try:
                importlib.import_module(pkg)
            except Exception:
                sys.modules[pkg] = _inert(pkg)
                added.append(pkg)
        spec = importlib.util.spec_from_file_location(alias_name + "._reference", path)
        mod = importlib.util.module_from_spec(spec)
        try:
            spec.loader.exec_module(mod)
            res = (mod, path)
        except Exception as e:                             # a third-party package that is really needed, e.g. torchvision
            res = (None, f"{path}: {type(e).__name__}: {e}")
        finally:
            for pkg in added:
                sys.modules.pop(pkg, None)
    _cache[relpath] = res
    return res


def lazy_reference_getattr(alias_name, relpath, optional_packages=()):
    """Module-level ``__getattr__`` for an alias module: names this repo does not provide are looked up in the reference's
    module of the same path, loaded on first use."""
    def __getattr__(name):
        if name.startswith("__"):
            raise AttributeError(name)
        mod, why = reference_module(alias_name, relpath, optional_packages)
        if mod is not None and hasattr(mod, name):
            return getattr(mod, name)
        raise AttributeError(f"{alias_name}.{name} is not part of the MI355X render path and the reference checkout behind "
                             f"this repo on sys.path does not provide it ({why})")
    return __getattr__


def _forward_returns_im_out(path):
    """Does ``UNet.forward`` in the module at `path` RETURN a dict with the key 'im_out' (src/READ/models/unet.py:280)?  Read off the
    syntax tree — a comment or a string elsewhere in the file that mentions 'im_out' decides nothing."""
    import ast
    try:
        with open(path, "r", errors="replace") as f:
            tree = ast.parse(f.read())
    except (OSError, SyntaxError, ValueError):
        return False
    for cls in (n for n in ast.walk(tree) if isinstance(n, ast.ClassDef) and n.name == "UNet"):
        for fn in (n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name == "forward"):
            for ret in (n for n in ast.walk(fn) if isinstance(n, ast.Return) and isinstance(n.value, ast.Dict)):
                if any(isinstance(k, ast.Constant) and k.value == "im_out" for k in ret.value.keys):
                    return True
    return False


_convention_path = None


def result_convention():
    """'dict' when the net's result is ``{'im_out': tensor}`` (the reference's ``src`` tree), 'tensor' otherwise (root tree,
    or no checkout).  Order: ``set_result_convention`` / the environment variable ``READ_AMD_RESULT`` (``tensor`` | ``dict``),
    else what ``UNet.forward`` of the ``READ/models/unet.py`` behind this repo RETURNS (its syntax tree, not its text).  The
    detection is remembered per resolved file: when ``sys.path`` changes and another tree moves behind the alias package, the
    next call looks again."""
    global _convention, _convention_path
    env = os.environ.get("READ_AMD_RESULT", "").strip().lower()
    if _forced is not None:
        return _forced
    if env in ("tensor", "dict"):
        return env
    path = reference_file(os.path.join("READ", "models", "unet.py"))
    if _convention is None or path != _convention_path:
        _convention_path = path
        _convention = "dict" if (path is not None and _forward_returns_im_out(path)) else "tensor"
    return _convention


_forced = None


def set_result_convention(value):
    """Force 'tensor' / 'dict' (None = detect again at the next call)."""
    global _convention, _forced
    if value not in (None, "tensor", "dict"):
        raise ValueError("result convention is 'tensor', 'dict' or None")
    _forced = value
    _convention = None
