"""CPU restatement (torch fp32) of the reference's floating-point path.

TEST INFRASTRUCTURE ONLY -- the checker, never the product.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

Each class / function cites the reference lines (relative to /root/reference) it
follows.  Parameter names reproduce the reference's ``state_dict`` keys (SURVEY
Appendix C) so one weight set drives the reference modules, this oracle and the
HIP path.  Third-party ops the reference takes from pytorch3d are defined by
oracle/pcc_oracle.c (orc_knn etc.; PARITY UNPINNED for tie order -- pytorch3d is
absent from the image).

Pins: tests/test_oracle_golden.py compares these restatements with outputs of
the reference's own pn_kit.py / AE.py modules captured by tests/golden/make_golden.py.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import cport

OCTREE_BPP_DICT = {1024: 0.07, 512: 0.125, 256: 0.25, 128: 0.5, 64: 1.0}  # pn_kit.py:17-23


# --------------------------------------------------------------------------- pytorch3d-named ops
def knn_points(p1, p2, K, return_nn=True):
    """pytorch3d.ops.knn_points as the oracle defines it: (dists, idx, nn)."""
    B = p1.shape[0]
    ds, ids = [], []
    for b in range(B):
        d, i = cport.knn(p1[b].detach().cpu().numpy(), p2[b].detach().cpu().numpy(), K)
        ds.append(torch.from_numpy(d))
        ids.append(torch.from_numpy(i))
    dists, idx = torch.stack(ds), torch.stack(ids)
    nn_ = None
    if return_nn:
        nn_ = torch.stack([p2[b][idx[b]] for b in range(B)])  # (B,M,K,3); a fresh tensor (callers do -= on it)
    return dists, idx, nn_


def chamfer_distance(x, y):
    """pytorch3d.loss.chamfer_distance defaults (AE.py:67, eval.py:204): squared
    distances, mean over points, both directions summed, mean over batch."""
    B = x.shape[0]
    tot = 0.0
    for b in range(B):
        dxy, _ = cport.nn_dist(x[b].detach().cpu().numpy(), y[b].detach().cpu().numpy())
        dyx, _ = cport.nn_dist(y[b].detach().cpu().numpy(), x[b].detach().cpu().numpy())
        tot += float(np.mean(dxy.astype(np.float64)) + np.mean(dyx.astype(np.float64)))
    return tot / B, None


# --------------------------------------------------------------------------- pn_kit.py pieces
def normalize(pc, margin=0.01):
    """pn_kit.normalize (pn_kit.py:47-60).  pc (1,N,3)."""
    x, y, z = pc[0, :, 0], pc[0, :, 1], pc[0, :, 2]
    center = torch.stack([(x.max() + x.min()) / 2, (y.max() + y.min()) / 2, (z.max() + z.min()) / 2])
    longest = torch.stack([x.max() - x.min(), y.max() - y.min(), z.max() - z.min()]).max()
    pc = pc - center
    pc = pc * (1 - margin) / longest
    pc = pc + 0.5
    return pc, center, longest


def denormalize(pc, center, longest, margin=0.01):
    """pn_kit.denormalize (pn_kit.py:62-66)."""
    pc = pc - 0.5
    pc = pc * longest / (1 - margin)
    return pc + center


def farthest_point_sample(xyz, npoint, start_idx):
    """pn_kit.farthest_point_sample_batch (pn_kit.py:309-330) with explicit start."""
    out = [cport.fps(xyz[b].cpu().numpy(), npoint, int(start_idx[b])) for b in range(xyz.shape[0])]
    return torch.from_numpy(np.stack(out))


def index_points(points, idx):
    """pn_kit.index_points (pn_kit.py:332-360)."""
    B = points.shape[0]
    bi = torch.arange(B).view([B] + [1] * (idx.dim() - 1)).expand_as(idx)
    return points[bi, idx.long(), :]


def pmf_to_cdf(pmf):
    """pn_kit.pmf_to_cdf (pn_kit.py:452-461)."""
    cdf = pmf.cumsum(dim=-1)
    z = torch.zeros(pmf.shape[:-1] + (1,), dtype=pmf.dtype)
    return torch.cat([z, cdf], dim=-1).clamp(max=1.0)


def cdf_float_to_int(cdf_float):
    """torchac._convert_to_int_and_normalize (torchac 0.9.3, needs_normalization=True):
    round(cdf * (2^16 - (Lp-1))) + arange(Lp), wrapped to 16 bits.  PARITY UNPINNED."""
    Lp = cdf_float.shape[-1]
    c = (cdf_float.float() * float(65536 - (Lp - 1))).round().to(torch.int64) + torch.arange(Lp)
    return (c & 0xFFFF).to(torch.int32).numpy()


def estimate_bits_from_pmf(pmf, sym):
    """pn_kit.estimate_bits_from_pmf (pn_kit.py:439-450)."""
    L = pmf.shape[-1]
    p = torch.gather(pmf.reshape(-1, L), 1, sym.reshape(-1, 1).long())
    return torch.sum(-torch.log2(p.clamp(min=1e-3)))


def _conv_stack(name_is_seq, chans, relu):
    mods = nn.ModuleList()
    for i in range(len(chans) - 1):
        layers = [nn.Conv2d(chans[i], chans[i + 1], 1)]
        if relu[i]:
            layers.append(nn.ReLU())
        mods.append(nn.Sequential(*layers))
    return mods


class PointNet(nn.Module):
    """pn_kit.PointNet (pn_kit.py:98-144), bn=False: 1x1 conv stack, max over points."""

    def __init__(self, in_channel, mlps, relu):
        super().__init__()
        self.mlp_Modules = _conv_stack(True, [in_channel] + list(mlps), relu)

    def forward(self, points):  # (B,C,N)
        x = points.unsqueeze(-1)
        for m in self.mlp_Modules:
            x = m(x)
        return torch.max(x, 2)[0].squeeze(-1)


class MLP(nn.Module):
    """pn_kit.MLP (pn_kit.py:263-305), bn=False."""

    def __init__(self, in_channel, mlps, relu):
        super().__init__()
        self.mlp_Modules = _conv_stack(True, [in_channel] + list(mlps), relu)

    def forward(self, points):
        x = points.unsqueeze(-1)
        for m in self.mlp_Modules:
            x = m(x)
        return x.squeeze(-1)


class SetAbstraction(nn.Module):
    """pn_kit.SetAbstraction (pn_kit.py:146-211), bn=False, npoint == N branch (:181-182)
    plus the FPS branch (:184) with start index 0 per batch element for determinism."""

    def __init__(self, npoint, K, in_channel, mlp, finalRelu=True):
        super().__init__()
        self.npoint, self.K, self.finalRelu = npoint, K, finalRelu
        self.conv0 = nn.Conv2d(in_channel + 3, mlp[0], 1)
        self.conv1 = nn.Conv2d(mlp[0], mlp[1], 1)
        self.conv2 = nn.Conv2d(mlp[1], mlp[2], 1)

    def forward(self, xyz):  # (B,3,N)
        xyz = xyz.permute(0, 2, 1)
        B, N, Cc = xyz.shape
        S = self.npoint
        if S == N:
            new_xyz = xyz
        else:
            new_xyz = index_points(xyz, farthest_point_sample(xyz, S, [0] * B))
        _, _, grouped = knn_points(new_xyz, xyz, K=self.K, return_nn=True)
        grouped = grouped - new_xyz.reshape(B, S, 1, Cc)
        g = grouped.permute(0, 3, 2, 1)  # (B,3,K,S)
        g = F.relu(self.conv0(g))
        g = F.relu(self.conv1(g))
        g = self.conv2(g)
        if self.finalRelu:
            g = F.relu(g)
        return new_xyz.permute(0, 2, 1), torch.max(g, 2)[0]


class AE(nn.Module):
    """AE.AE (AE.py:12-55)."""

    def __init__(self, K, k, d, L):
        super().__init__()
        self.sa = SetAbstraction(npoint=K, K=16, in_channel=0, mlp=[32, 64, 128])
        self.pn = PointNet(3 + 128, [128, 256, 512, d], [True, True, True, False])
        self.inv_pool = nn.Sequential(nn.Linear(d, 256), nn.ReLU(), nn.Linear(256, 1024), nn.ReLU(),
                                      nn.Linear(1024, k * 128), nn.ReLU())
        self.inv_mlp = MLP(d + 128, [128, 64, 32, 3], [True, True, True, False])
        self.K, self.k, self.L, self.d = K, k, L, d

    def encode(self, xyz):  # (BS,K,3) -> latent (BS,d) before quantisation (AE.py:37-44)
        xyz = xyz.transpose(2, 1)
        _, feat = self.sa(xyz)
        latent = self.pn(torch.cat((xyz, feat), dim=1))
        spread = self.L - 0.2
        return torch.sigmoid(latent) * spread - spread / 2

    def decode(self, latent_q):  # (BS,d) -> (BS,k,3)  (AE.py:48-53; decompress.py:97-102)
        BS = latent_q.shape[0]
        lin = self.inv_pool(latent_q).view(BS, -1, self.k)
        rep = latent_q.unsqueeze(-1).repeat((1, 1, self.k))
        return self.inv_mlp(torch.cat((lin, rep), dim=1)).transpose(2, 1)

    def forward(self, xyz):
        latent = self.encode(xyz)
        q = latent.round()
        return self.decode(q), latent, q


class ConditionalProbabilityModel(nn.Module):
    """AE.ConditionalProbabilityModel (AE.py:87-123)."""

    def __init__(self, L, d):
        super().__init__()
        self.L, self.d = L, d
        self.model_pn = PointNet(3, [64, 128, 256], [True, True, True])
        self.model_mlp = nn.Sequential(nn.Conv2d(3 + 256, 512, 1), nn.ReLU(), nn.Conv2d(512, 512, 1), nn.ReLU(),
                                       nn.Conv2d(512, d * L, 1))

    def forward(self, sampled_xyz):  # (B,S,3)
        B, S, _ = sampled_xyz.shape
        feature = self.model_pn(sampled_xyz.transpose(1, 2))
        mlp_input = torch.cat((sampled_xyz, feature.repeat((1, S)).view(B, S, -1)), dim=2)
        out = self.model_mlp(mlp_input.unsqueeze(-1).transpose(1, 2))
        out = out.transpose(1, 2).reshape(B, S, self.d, self.L)
        return F.softmax(out, dim=3)


# --------------------------------------------------------------------------- deterministic weights
def seeded_state_dict(module, seed, gain=1.0, last_gain=None):
    """Deterministic, torch-RNG-independent weights keyed by state_dict name order.

    Every tensor is filled from numpy default_rng(seed) uniform(-b,b) with
    b = gain/sqrt(fan_in) (biases use the same bound).  ``last_gain`` maps a key
    substring -> multiplier (used to spread the quantiser's symbols over -3..3).
    """
    rng = np.random.default_rng(seed)
    sd = {}
    for k, v in module.state_dict().items():
        shape = tuple(v.shape)
        if len(shape) == 0:                     # e.g. BatchNorm.num_batches_tracked
            sd[k] = v.clone()
            continue
        fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else int(shape[0])
        if k.endswith("bias"):
            w = module.state_dict()[k[:-4] + "weight"]
            fan_in = int(np.prod(tuple(w.shape)[1:]))
        b = gain / np.sqrt(max(fan_in, 1))
        a = rng.uniform(-b, b, size=shape).astype(np.float32)
        if last_gain:
            for sub, g in last_gain.items():
                if sub in k:
                    a = a * np.float32(g)
        sd[k] = torch.from_numpy(a)
    return sd
