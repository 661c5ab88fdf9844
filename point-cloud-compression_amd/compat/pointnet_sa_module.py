"""Drop-in for the reference's ``pointnet_sa_module`` (pointnet_sa_module.py)."""
import pn_kit  # noqa: F401
from pccx import ops
from pccx.families import PointnetSAModule  # noqa: F401


class PointnetPPOps:                                              # pointnet_sa_module.py:8-34
    @staticmethod
    def furthest_point_sample(xyz, npoint):
        return ops.sample_farthest_points(xyz, npoint)[1]

    @staticmethod
    def ball_query(radius, nsample, xyz, new_xyz):
        return ops.ball_query(new_xyz, xyz, nsample, radius)

    @staticmethod
    def group_points(features, idx):
        if hasattr(idx, "idx"):
            idx = idx.idx
        return ops.knn_gather(features, idx.clamp(min=0))

    @staticmethod
    def knn_point(k, xyz, new_xyz):
        r = ops.knn_points(new_xyz, xyz, k)
        return r.dists, r.idx
