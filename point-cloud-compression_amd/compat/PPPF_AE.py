"""Drop-in for the reference's ``PPPF_AE`` module (PPPF_AE.py)."""
import torch

import pn_kit  # noqa: F401
from pccx import ops
from pccx.families import FoldingNet, PointNetPP, PPPF_AE  # noqa: F401

AE = PPPF_AE                                                      # PPPF_AE.py:230-232


class get_loss(torch.nn.Module):                                  # PPPF_AE.py:153-177 (forward value only)
    def forward(self, pc_pred, pc_target, fbpp, λ):
        d, _ = ops.chamfer_distance(pc_pred, pc_target)
        r = fbpp.mean() if isinstance(fbpp, torch.Tensor) and fbpp.ndim > 0 else fbpp
        return d + λ * r
