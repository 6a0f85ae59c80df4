"""4D control-map renderer (reference: inference/rendering_4D_control_maps.py)."""
from . import control_maps  # noqa: F401
