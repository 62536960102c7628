"""Alias of READ/gl/dataset.py:39-82 (the input-format DSL)."""
from read_amd.render import parse_input_string  # noqa: F401
