"""Alias of READ/gl/programs.py: ``NNScene`` is the GL-free scene of read_amd (same setters and mode constants), so
``DynamicDataset.load`` (READ/datasets/dynamic.py:172-179), ``viewer.py`` and ``READ.gl.dataset.parse_input_string`` keep
working without an OpenGL context."""
from read_amd.render import Scene as NNScene  # noqa: F401
