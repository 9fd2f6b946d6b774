"""Model registry — the reference's plug-in boundary: `get_model(args)` maps `args.model` to a MODULE
exposing `Model` (reference model_dict.py).  The structured-mesh-2D family is the hot path;
the irregular-mesh family is its §8(f)-2 widening; the other two reference families are not built."""
from .model import Transolver_Irregular_Mesh, Transolver_Structured_Mesh_2D

REGISTRY = {
    "Transolver_Structured_Mesh_2D": Transolver_Structured_Mesh_2D,     # exp_ns / exp_darcy / exp_airfoil / exp_pipe / exp_plas
    "Transolver_Irregular_Mesh": Transolver_Irregular_Mesh,             # exp_elas
}
NOT_BUILT = ("Transolver_Structured_Mesh_3D", "Transolver_Structured_Mesh2D_Encoder")


def get_model(args):
    name = args.model
    if name in NOT_BUILT:
        raise KeyError(f"{name}: not part of the MI355X-native build (DESIGN.md §7)")
    return REGISTRY[name]     # KeyError for unknown names, like the reference's dict lookup
