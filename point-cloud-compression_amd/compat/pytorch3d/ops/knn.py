from . import knn_gather, knn_points  # noqa: F401
from pccx.ops import KNN as _KNN  # noqa: F401
