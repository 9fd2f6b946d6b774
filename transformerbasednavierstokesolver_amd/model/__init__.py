"""Model modules mirroring the reference's `model/` package (structured-mesh-2D and irregular-mesh families)."""
from . import Transolver_Structured_Mesh_2D, Transolver_Irregular_Mesh, SOL_Transolver_Structured_Mesh_2D, Physics_Attention  # noqa: F401
