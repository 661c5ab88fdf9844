import os
import sys

_PKG = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if _PKG not in sys.path:
    sys.path.insert(0, _PKG)

from pccx.ops import ball_query as _bq, knn_gather, knn_points, sample_farthest_points  # noqa: E402,F401


def ball_query(p1, p2, K=500, radius=0.2):
    return _bq(p1, p2, K, radius)
