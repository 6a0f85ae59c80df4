"""Pre-processing on the engine: the per-object 3D Gaussian fit (reference: inference/fit_3D_gaussian.py) and the 4D control-map
renderer (reference: inference/rendering_4D_control_maps.py)."""
from . import control_maps, gaussian_fit  # noqa: F401
