"""Drop-in for the reference's ``AE`` module (AE.py): same class names, constructor arguments and
state_dict keys; forward passes run on libpccx.so."""
import torch

import pn_kit  # noqa: F401  (sets sys.path)
from pccx import ops
from pccx.models import AE, ConditionalProbabilityModel  # noqa: F401


class get_loss(torch.nn.Module):                                  # AE.py:57-70
    """Rate-distortion loss d + lambda * r.  The Chamfer term is differentiable w.r.t. both clouds
    (ops.chamfer_distance: HIP forward + pccx_chamfer_grad backward); the models of this module are
    inference-only (their forwards do not record a graph), so training goes through pccx.train."""

    def forward(self, pc_pred, pc_target, fbpp, λ):
        d, _ = ops.chamfer_distance(pc_pred, pc_target)
        return d + λ * fbpp


class STEQuantize:                                                # AE.py:72-85
    """apply(x) = x.round() with the straight-through gradient of AE.py:83-85 (ops.ste_round)."""
    apply = staticmethod(ops.ste_round)
