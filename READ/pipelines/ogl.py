"""Alias of READ/pipelines/ogl.py: ``TexturePipeline`` (:58-154) is the MI355X one; the pipelines of the other model
families (``Pix2PixPipeline``, ``RGBTexturePipeline``) and the module's helpers keep coming from the reference checkout
behind this repo on ``sys.path``."""
from read_amd._alias import lazy_reference_getattr
from read_amd.pipeline import TexturePipeline, TextureOptimizerClass  # noqa: F401

__getattr__ = lazy_reference_getattr(__name__, "READ/pipelines/ogl.py", optional_packages=("cv2", "imageio", "huepy"))
