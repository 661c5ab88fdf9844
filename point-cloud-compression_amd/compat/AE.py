"""Drop-in for the reference's ``AE`` module (AE.py): same class names, constructor arguments and
state_dict keys; forward passes run on libpccx.so."""
import torch

import pn_kit  # noqa: F401  (sets sys.path)
from pccx import ops
from pccx.models import AE, ConditionalProbabilityModel  # noqa: F401


class get_loss(torch.nn.Module):                                  # AE.py:57-70 (forward value only)
    def forward(self, pc_pred, pc_target, fbpp, λ):
        d, _ = ops.chamfer_distance(pc_pred, pc_target)
        return d + λ * fbpp


class STEQuantize:                                                # AE.py:72-85 (forward)
    @staticmethod
    def apply(x):
        return x.round()
