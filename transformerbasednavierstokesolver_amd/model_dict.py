"""Model registry — the reference's plug-in boundary (model_dict.py:4-11): `get_model(args)` returns a
MODULE that exposes `Model`.  The structured-mesh-2D family is the hot path; the irregular-mesh family is §8(f)-2."""
from .model import Transolver_Structured_Mesh_2D, Transolver_Irregular_Mesh

_OUT_OF_SCOPE = ('Transolver_Structured_Mesh_3D', 'Transolver_Structured_Mesh2D_Encoder')


def get_model(args):
    model_dict = {
        'Transolver_Irregular_Mesh': Transolver_Irregular_Mesh,          # SURVEY 8(f)-2 (exp_elas.py)
        'Transolver_Structured_Mesh_2D': Transolver_Structured_Mesh_2D,
    }
    if args.model in _OUT_OF_SCOPE:
        raise KeyError(f"{args.model}: not part of the MI355X-native hot path (see DESIGN.md, out of scope)")
    return model_dict[args.model]
