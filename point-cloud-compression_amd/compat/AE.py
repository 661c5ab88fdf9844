"""Drop-in for the reference's ``AE`` module (AE.py): same class names, constructor arguments and
state_dict keys; forward passes run on libpccx.so."""
import torch

import pn_kit  # noqa: F401  (sets sys.path)
from pccx import ops
from pccx.models import AE, ConditionalProbabilityModel  # noqa: F401


class get_loss(torch.nn.Module):                                  # AE.py:57-70
    """Rate-distortion loss d + lambda * r.  The Chamfer term is differentiable w.r.t. both clouds
    (ops.chamfer_distance: HIP forward + pccx_chamfer_grad backward).  The forwards of this module's models are
    the codec's inference kernels (they record no graph); the training step of train.py:156-245 evaluates the SAME
    parameters layer by layer through autograd Functions in pccx.train_ipdae (cli/train.py keeps train.py's flags)."""

    def forward(self, pc_pred, pc_target, fbpp, λ):
        d, _ = ops.chamfer_distance(pc_pred, pc_target)
        return d + λ * fbpp


class STEQuantize:                                                # AE.py:72-85
    """apply(x) = x.round() with the straight-through gradient of AE.py:83-85 (ops.ste_round)."""
    apply = staticmethod(ops.ste_round)
