from read_amd.pipeline import Pipeline, load_pipeline, save_pipeline  # noqa: F401
