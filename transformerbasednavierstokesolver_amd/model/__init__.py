"""Model modules mirroring the reference's `model/` package for the structured-mesh-2D path."""
from . import Transolver_Structured_Mesh_2D, SOL_Transolver_Structured_Mesh_2D, Physics_Attention  # noqa: F401
