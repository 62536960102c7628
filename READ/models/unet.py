from read_amd.unet import UNet  # noqa: F401
