"""Minimal 3DGS trainer surface that `model/diffusionGS.py` drives (the FSGS submodule is absent from the
reference container): cameras, Gaussian parameters, `render_view` and the HOT LOOP A
optimisation loop on the HIP rasteriser."""
from .trainer import Camera, GaussianModel, GSTrainer, OptimizationParams  # noqa: F401
