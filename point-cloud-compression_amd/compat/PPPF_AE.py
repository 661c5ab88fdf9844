"""Drop-in for the reference's ``PPPF_AE`` module (PPPF_AE.py)."""
import pn_kit  # noqa: F401
from pccx.families import FoldingNet, PointNetPP, PPPF_AE  # noqa: F401

AE = PPPF_AE                                                      # PPPF_AE.py:230-232
