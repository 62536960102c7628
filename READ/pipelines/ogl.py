from read_amd.pipeline import TexturePipeline, TextureOptimizerClass  # noqa: F401
