"""Alias package: lets the reference's dotted paths (``--pipeline READ.pipelines.ogl.TexturePipeline``,
``from READ.models.unet import UNet`` ...) resolve to the MI355X implementation in ``read_amd``.
Only the render-path modules exist here.  ``__path__`` is extended over every other ``READ`` directory on ``sys.path``
(``pkgutil.extend_path``), so with this repo IN FRONT of the reference checkout everything else of READ (datasets,
criterions, utils, viewer helpers) keeps coming from the reference (tests/test_read_alias.py)."""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)   # the rest of this package keeps resolving to the reference checkout
