"""Alias of READ/models/unet.py:121-285 (root tree: returns the image) and src/READ/models/unet.py (returns
``{'im_out': image}``, :280) — ``read_amd.unet.UNet`` follows whichever tree sits behind this repo on ``sys.path``
(``read_amd._alias.result_convention``)."""
from read_amd.unet import UNet  # noqa: F401
