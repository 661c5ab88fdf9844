"""CPU restatement (torch fp32, eval mode) of the reference's other two model families:

  * PPPF_AE.py + pointnet_sa_module.py  (BASELINE configs[2])
  * pppe_pcd_ae.py:556-917 forward      (SURVEY 8a row a21;        BASELINE configs[4])

TEST INFRASTRUCTURE ONLY -- the checker, never the product.

Module / parameter names reproduce the reference's state_dict keys.  pytorch3d's
sample_farthest_points / ball_query / knn_points / knn_gather are the oracle's definitions
(oracle/pcc_oracle.c; PARITY UNPINNED for tie order and padding, pytorch3d is absent).
BatchNorm runs in eval mode (running statistics), which is how compress-side inference uses it.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import cport, ref_model


# ---- pytorch3d-named ops (oracle definitions) -------------------------------------------------
def sample_farthest_points(xyz, K):
    """Start index 0 (pointnet_sa_module.py:12 relies on the default)."""
    idx = torch.from_numpy(np.stack([cport.fps(xyz[b].numpy(), K, 0) for b in range(xyz.shape[0])]))
    return ref_model.index_points(xyz, idx), idx


def ball_query(p1, p2, K, radius):
    ds, ids = [], []
    for b in range(p1.shape[0]):
        d, i = cport.ball_query(p1[b].numpy(), p2[b].numpy(), K, radius)
        ds.append(torch.from_numpy(d)); ids.append(torch.from_numpy(i))
    return torch.stack(ds), torch.stack(ids), None


def knn_gather(x, idx):
    return ref_model.index_points(x, idx)


# ---- pointnet_sa_module.py:38-93 ----------------------------------------------------------------
class PointnetSAModule(nn.Module):
    def __init__(self, npoint, radius, nsample, mlp, use_xyz=True, in_channels=0):
        super().__init__()
        self.npoint, self.radius, self.nsample, self.use_xyz = npoint, radius, nsample, use_xyz
        last = in_channels + (3 if use_xyz else 0)
        layers = []
        for out in mlp:
            layers += [nn.Conv2d(last, out, 1), nn.BatchNorm2d(out), nn.ReLU(inplace=True)]
            last = out
        self.mlp = nn.Sequential(*layers)

    def forward(self, xyz, features=None):
        _, fps_idx = sample_farthest_points(xyz, self.npoint)                       # :66
        fps_idx = fps_idx.clamp(min=0)
        new_xyz = torch.gather(xyz, 1, fps_idx.unsqueeze(-1).expand(-1, -1, 3))     # :68
        _, idx, _ = ball_query(new_xyz, xyz, K=self.nsample, radius=self.radius)    # :71
        idx = idx.clamp(min=0)                                                      # :27
        grouped = None
        if features is not None:
            grouped = knn_gather(features.permute(0, 2, 1), idx)                    # :75-76
        if self.use_xyz:
            gx = knn_gather(xyz, idx)                                               # :81 (not centred)
            grouped = torch.cat([grouped, gx], dim=-1) if grouped is not None else gx
        out = self.mlp(grouped.permute(0, 3, 1, 2))                                 # :89-90
        return new_xyz, torch.max(out, 3)[0]                                        # :91


# ---- PPPF_AE.py:9-150 -----------------------------------------------------------------------------
class PointNetPP(nn.Module):
    def __init__(self, points=512, sa1_mlp=(64, 64, 128), sa2_mlp=(128, 128, 128, 256), sa3_mlp=(256, 256, 512),
                 feature_dim=1024):
        super().__init__()
        self.sa1 = PointnetSAModule(points, 0.2, 32, [3] + list(sa1_mlp), True, 0)          # first conv is 3->3 (:30)
        self.sa2 = PointnetSAModule(128, 0.4, 64, list(sa2_mlp), True, 128)
        self.sa3 = PointnetSAModule(32, 0.8, 128, list(sa3_mlp) + [feature_dim], True, 256)

    def forward(self, xyz, features=None):
        xyz, features = self.sa1(xyz, features)
        xyz, features = self.sa2(xyz, features)
        xyz, features = self.sa3(xyz, features)
        return xyz, torch.max(features, dim=2)[0]


class FoldingNet(nn.Module):
    def __init__(self, points=512, grid_size=45, feature_dim=1024):
        super().__init__()
        self.grid_size, self.num_points = grid_size, grid_size * grid_size
        self.mlp1 = nn.Sequential(nn.Conv1d(feature_dim + 2, points, 1), nn.ReLU(), nn.Conv1d(points, points, 1), nn.ReLU(),
                                  nn.Conv1d(points, 3, 1))
        self.mlp2 = nn.Sequential(nn.Conv1d(feature_dim + 3, 128, 1), nn.ReLU(), nn.Conv1d(128, 128, 1), nn.ReLU(),
                                  nn.Conv1d(128, 3, 1))

    def build_grid(self, B):
        x = torch.linspace(-1, 1, self.grid_size)
        gx, gy = torch.meshgrid(x, x, indexing="ij")
        return torch.stack([gx, gy], dim=-1).reshape(-1, 2).unsqueeze(0).repeat(B, 1, 1)

    def forward(self, latent):
        B = latent.size(0)
        grid = self.build_grid(B)
        rep = latent.unsqueeze(1).repeat(1, self.num_points, 1)
        coarse = self.mlp1(torch.cat([grid, rep], dim=-1).transpose(2, 1))
        fine = self.mlp2(torch.cat([coarse, rep.transpose(2, 1)], dim=1))
        return fine.transpose(2, 1)


class PPPF_AE(nn.Module):
    def __init__(self, K=512, k=0, d=16, L=7, dim=1024):
        super().__init__()
        self.L = L
        self.encoder = PointNetPP(points=K, feature_dim=dim)
        self.decoder = FoldingNet(points=K, grid_size=d)
        self.enc_proj = nn.Linear(dim, d)
        self.dec_proj = nn.Linear(d, dim)

    def forward(self, xyz):
        _, latent = self.encoder(xyz)
        spread = self.L - 0.2
        latent = torch.sigmoid(latent) * spread - spread / 2
        z = self.enc_proj(latent)
        q = z.round()
        return self.decoder(self.dec_proj(q)), latent, q, z


# ---- pppe_pcd_ae.py:556-877 -----------------------------------------------------------------------
def _c2(in_c, out_c):
    return nn.Sequential(nn.Conv2d(in_c, out_c, 1, bias=False), nn.BatchNorm2d(out_c), nn.ReLU(inplace=True))


class PointNetSetAbstraction(nn.Module):
    def __init__(self, npoint, K, in_channel, mlp):
        super().__init__()
        self.npoint, self.K = npoint, K
        last = in_channel + 3
        layers = []
        for out in mlp:
            layers.append(_c2(last, out))
            last = out
        self.mlp_stack = nn.ModuleList(layers)

    def forward(self, xyz, points, start):
        B, N, _ = xyz.shape
        S = self.npoint
        if S == N:
            new_xyz = xyz
        else:
            new_xyz = ref_model.index_points(xyz, ref_model.farthest_point_sample(xyz, S, start))   # :596-597
        _, idx, grouped = ref_model.knn_points(new_xyz, xyz, K=self.K, return_nn=True)              # :599
        grouped = grouped - new_xyz.view(B, S, 1, 3)
        if points is not None:
            grouped = torch.cat([grouped, ref_model.index_points(points.permute(0, 2, 1), idx)], dim=-1)   # xyz first (:606)
        g = grouped.permute(0, 3, 2, 1).contiguous()
        for layer in self.mlp_stack:
            g = layer(g)
        return new_xyz, torch.max(g, dim=2)[0]


class PointNetSetAbstractionMSG(nn.Module):
    def __init__(self, npoint, scales, in_channel):
        super().__init__()
        self.branches = nn.ModuleList([PointNetSetAbstraction(npoint, s["K"], in_channel, s["mlp"]) for s in scales])

    def forward(self, xyz, points, starts):
        outs, new_xyz = [], None
        for b, st in zip(self.branches, starts):       # each branch draws its own FPS start (:624-632)
            new_xyz, p = b(xyz, points, st)
            outs.append(p)
        return new_xyz, torch.cat(outs, dim=1)


class PointNet2EncoderFull(nn.Module):
    def __init__(self, latent_dim=256):
        super().__init__()
        self.sa_modules = nn.ModuleList([
            PointNetSetAbstractionMSG(512, [{"K": 16, "mlp": [32, 32, 64]}, {"K": 32, "mlp": [64, 64, 128]}], 0),
            PointNetSetAbstraction(128, 32, 64 + 128, [128, 128, 256]),
            PointNetSetAbstraction(32, 32, 256, [256, 256, 512])])
        self.global_conv = nn.Sequential(nn.Conv1d(512, 512, 1, bias=False), nn.BatchNorm1d(512), nn.ReLU(inplace=True),
                                         nn.Conv1d(512, latent_dim, 1))

    def forward(self, x, starts):
        """starts = [[s_msg_branch0, s_msg_branch1], s_sa2, s_sa3], each a length-B list of FPS start indices."""
        xyz, points = self.sa_modules[0](x, None, starts[0])
        xyz, points = self.sa_modules[1](xyz, points, starts[1])
        xyz, points = self.sa_modules[2](xyz, points, starts[2])
        gf = torch.max(points, dim=2)[0]
        return self.global_conv(gf.unsqueeze(-1)).squeeze(-1), gf


class PCNDecoderSmall(nn.Module):
    def __init__(self, latent_dim=256, coarse_points=512, final_points=8192):
        super().__init__()
        self.fc_coarse = nn.Sequential(nn.Linear(latent_dim, 512), nn.ReLU(), nn.Linear(512, coarse_points * 3))
        self.expansion_mlp = nn.Sequential(nn.Linear(coarse_points * 3 + latent_dim, 1024), nn.ReLU(),
                                           nn.Linear(1024, final_points * 3))
        self.coarse_points, self.final_points = coarse_points, final_points

    def forward(self, latent):
        B = latent.size(0)
        coarse = self.fc_coarse(latent).view(B, self.coarse_points, 3)
        fine = self.expansion_mlp(torch.cat([coarse.view(B, -1), latent], dim=1)).view(B, self.final_points, 3)
        return coarse, fine


class PppeProb(nn.Module):
    """pppe_pcd_ae.ConditionalProbabilityModel (:740-802); only parameter shapes matter to PointCloudAE's keys."""

    def __init__(self, feature_dim=512, hidden_channels=128, latent_bins=16, latent_channels=3):
        super().__init__()
        self.cond_proj = nn.Sequential(nn.Linear(feature_dim, hidden_channels), nn.ReLU(), nn.Linear(hidden_channels, hidden_channels))
        self.combine = nn.Sequential(nn.Conv1d(latent_channels + hidden_channels, hidden_channels, 1), nn.ReLU(),
                                     nn.Conv1d(hidden_channels, hidden_channels, 1))
        self.mean_head = nn.Conv1d(hidden_channels, latent_channels, 1)
        self.scale_head = nn.Conv1d(hidden_channels, latent_channels, 1)
        self.pmf_head = nn.Conv1d(hidden_channels, latent_bins, 1)


class PointCloudAE(nn.Module):
    def __init__(self, latent_dim=64, latent_bins=16, npoints=8192):
        super().__init__()
        self.encoder = PointNet2EncoderFull(latent_dim=latent_dim)
        self.decoder = PCNDecoderSmall(latent_dim=latent_dim, coarse_points=512, final_points=npoints)
        self.prob = PppeProb(512, 128, latent_bins, latent_dim)
        self.latent_bins, self.latent_dim = latent_bins, latent_dim
        self.q_min, self.q_max = 0.0, latent_bins - 1.0

    def forward(self, x, starts):
        latent, cond = self.encoder(x, starts)                                     # :865
        xc = torch.clamp(latent, self.q_min, self.q_max)                           # quantize_st (:719-735)
        scaled = (xc - self.q_min) / (self.q_max - self.q_min + 1e-9) * (self.latent_bins - 1)
        y_q = torch.round(scaled).detach() + (scaled - scaled.detach())            # straight-through (:732)
        y_q = torch.clamp(y_q, 0, self.latent_bins - 1)
        y_deq = (y_q / (self.latent_bins - 1)) * (self.q_max - self.q_min) + self.q_min   # :873
        coarse, fine = self.decoder(y_deq)      # mean over N identical tiled copies == the value itself (:875)
        return coarse, fine, cond, y_q, latent


def seeded_with_bn(module, seed, gain=1.0):
    """seeded_state_dict plus sane BatchNorm statistics (running_var > 0, non-trivial mean / affine)."""
    sd = ref_model.seeded_state_dict(module, seed, gain=gain)
    rng = np.random.default_rng(seed + 1)
    for k, v in module.state_dict().items():
        if k.endswith("running_var"):
            sd[k] = torch.from_numpy(rng.uniform(0.5, 1.5, size=tuple(v.shape)).astype(np.float32))
        elif k.endswith("running_mean"):
            sd[k] = torch.from_numpy(rng.uniform(-0.2, 0.2, size=tuple(v.shape)).astype(np.float32))
        elif k.endswith("num_batches_tracked"):
            sd[k] = torch.tensor(1, dtype=torch.long)
        elif v.dim() == 1 and k.endswith("weight"):          # BatchNorm gamma (conv / linear weights are >= 2-D)
            sd[k] = torch.from_numpy(rng.uniform(0.8, 1.2, size=tuple(v.shape)).astype(np.float32))
        elif v.dim() == 1 and k.endswith("bias") and (k[:-4] + "running_mean") in module.state_dict():
            sd[k] = torch.from_numpy(rng.uniform(-0.1, 0.1, size=tuple(v.shape)).astype(np.float32))
    return sd
