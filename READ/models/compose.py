"""Alias of READ/models/compose.py.  ``NetAndTexture`` (compose.py:84-181) and ``ModelAndLoss`` (compose.py:12-32, imported by
train.py:26 and READ/utils/train.py:13) are the MI355X ones, ``MultiscaleNet`` / ``RGBTexture`` their torch-only siblings; every
other name of the reference module (``BoxFilter``, ``GaussianLayer`` ...) is looked up in the reference's own compose.py when
that checkout sits behind this repo on ``sys.path`` (its unused ``imageio`` / ``cv2`` imports need not be installed)."""
from read_amd._alias import lazy_reference_getattr
from read_amd.net_texture import ModelAndLoss, MultiscaleNet, NetAndTexture, RGBTexture  # noqa: F401

__getattr__ = lazy_reference_getattr(__name__, "READ/models/compose.py", optional_packages=("imageio", "cv2", "PIL"))
