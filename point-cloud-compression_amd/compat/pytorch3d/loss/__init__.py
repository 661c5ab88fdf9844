from ..ops import knn_points  # noqa: F401  (path setup)
from pccx.ops import chamfer_distance  # noqa: F401
