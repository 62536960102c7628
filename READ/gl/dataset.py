"""Alias of READ/gl/dataset.py: the input-format DSL, both directions (``parse_input_string`` :39-82,
``generate_input_string`` :85-122).  The mesh / GL dataset classes of that module stay with the reference."""
from read_amd._alias import lazy_reference_getattr
from read_amd.render import generate_input_string, parse_input_string  # noqa: F401

__getattr__ = lazy_reference_getattr(__name__, "READ/gl/dataset.py", optional_packages=("cv2", "glumpy", "trimesh"))
