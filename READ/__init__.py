"""Alias package: lets the reference's dotted paths (``--pipeline READ.pipelines.ogl.TexturePipeline``,
``from READ.models.unet import UNet`` ...) resolve to the MI355X implementation in ``read_amd``.
Only the render-path modules exist here; everything else of READ (datasets, criterions, viewer) is
out of scope and should keep coming from the reference checkout."""
