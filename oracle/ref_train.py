"""CPU restatement (torch autograd) of one iteration of train_pppe_pcd_ae.train_one_epoch
(train_pppe_pcd_ae.py:184-226) without the CUDA autocast / GradScaler branch: forward in train mode,
estimate_bits_per_point_conditional (pppe_pcd_ae.py:882-917), RateDistortionLoss (pppe_pcd_ae.py:807-838;
the script builds it as get_loss("chamfer"), train_pppe_pcd_ae.py:48 -- 'hybrid' and 'l1' are the class's
other two modes), backward, clip_grad_norm_(ae + prob, 1.0), Adam step.

TEST INFRASTRUCTURE ONLY.  Pinned by tests/golden/train_step.npz, two iterations of the reference's own
train_one_epoch (tests/golden/make_golden.py section 6).  Chamfer itself is evaluated by brute force
(pytorch3d is absent; its definition here is the documented one, PARITY UNPINNED).
"""
import torch
import torch.nn.functional as F


def chamfer_autograd(x, y):
    d = ((x[:, :, None, :] - y[:, None, :, :]) ** 2).sum(-1)
    return (d.min(2).values.mean(1) + d.min(1).values.mean(1)).mean()


def chamfer_distance(x, y, batch_reduction="mean", point_reduction="mean"):
    """pytorch3d.loss.chamfer_distance as the reference calls it (AE.py:67, pppe_pcd_ae.py:825-830):
    squared L2, both directions, mean over points then over the batch; returns (loss, None)."""
    assert batch_reduction == "mean" and point_reduction == "mean"
    return chamfer_autograd(x, y), None


def prob_forward(pr, y, cond):
    """pppe_pcd_ae.ConditionalProbabilityModel.forward (:774-802) on one column (all N are identical)."""
    c = pr.cond_proj(cond)
    x = torch.cat([y, c], dim=1).unsqueeze(-1)
    h = pr.combine(x)
    return F.softmax(pr.pmf_head(h), dim=1).clamp(min=1e-9)            # (B,K,1)


def train_step(model, opt, batch_x, starts, lam=1.0, grad_clip=1.0, chamfer_chunk=None, loss_type="chamfer"):
    model.train()
    opt.zero_grad()
    coarse, fine, cond, y_q, _ = model(batch_x, starts)
    with torch.no_grad():
        pmf = prob_forward(model.prob, y_q.detach(), cond.detach())
        idx0 = torch.clamp(y_q[:, 0].long(), 0, pmf.shape[1] - 1).view(-1, 1, 1)
        fbpp = (-torch.log2(torch.gather(pmf, 1, idx0).clamp(min=1e-9))).mean()
    if loss_type == "chamfer":                                           # pppe_pcd_ae.py:825-827
        dist = chamfer_autograd(fine, batch_x)
    elif loss_type == "l1":                                              # :828-829
        dist = F.smooth_l1_loss(fine, batch_x, reduction="mean")
    else:                                                                # :830-833, alpha = 0.7
        dist = 0.7 * chamfer_autograd(fine, batch_x) + 0.3 * F.smooth_l1_loss(fine, batch_x, reduction="mean")
    rate = torch.clamp(fbpp, min=0.0, max=100.0)
    loss = dist + lam * rate
    loss.backward()
    torch.nn.utils.clip_grad_norm_(list(model.parameters()), grad_clip)
    opt.step()
    return float(loss.detach()), float(dist.detach()), float(rate.detach())


# ------------------------------------------------------------------------------------------------
def ipdae_train_step(ae, prob, opt, batch_x, starts, lam, N0=1024, ALPHA=2, K=256, chamfer=None):
    """One iteration of train.train_one_epoch for --model AE (train.py:162-236) on CPU, fp32 (the contextlib.nullcontext branch of
    :175): ae / prob are oracle.ref_model's restatements of AE.AE / AE.ConditionalProbabilityModel, opt a torch.optim.Adam over both
    (train.py:131-134).  -> dict(loss, fbpp, bpp).  TEST INFRASTRUCTURE ONLY.  Pinned by tests/golden/ipdae_train_step.npz (two
    iterations of the reference's own train_one_epoch; tests/test_train_ipdae.py checks this function against it on CPU).
    chamfer: a callable (pred, target) -> scalar with autograd, default the brute-force definition above."""
    import numpy as np
    from . import cport, ref_model
    B, N, _ = batch_x.shape
    S = N * ALPHA // K
    ae.train(), prob.train()
    # pn_kit.normalize reads centre / longest side from pc[0] only (pn_kit.py:50-53) and applies them to the whole batch
    x, _, _ = ref_model.normalize(batch_x, margin=0.01)                                            # :171
    opt.zero_grad()                                                                                # :173
    sampled = ref_model.index_points(x, ref_model.farthest_point_sample(x, S, list(starts)))       # :178
    codes, sampled_bits = cport.encode_sampled_np(sampled.numpy(), 1, N, ref_model.OCTREE_BPP_DICT[K])   # :183
    rec = torch.from_numpy(np.asarray(cport.decode_sampled_np(codes, 1, "reference"), dtype=np.float32))   # :184
    _, _, grouped = ref_model.knn_points(rec, x, K=K, return_nn=True)                               # :192-194
    grouped = grouped - rec.view(B, S, 1, 3)                                                       # :195
    scale = (N / N0) ** (1 / 3)
    x_patches = grouped.view(B * S, K, 3) * scale                                                  # :196-199
    latent = ae.encode(x_patches)                                                                  # AE.py:37-44
    q = latent + (latent.round() - latent).detach()                                                # AE.py:45, STEQuantize (:72-85)
    patches_pred = ae.decode(q) / scale                                                            # AE.py:48-53, train.py:201
    pmf = prob(rec)                                                                                # :204
    sym = (q.detach().view(B, S, ae.d) + ae.L // 2).long().clamp(0, ae.L - 1)                      # :205-206
    feature_bits = ref_model.estimate_bits_from_pmf(pmf, sym) / (B * N)                            # :208
    bpp = (sampled_bits + feature_bits) / (B * N)                                                  # :211
    fbpp = feature_bits / (B * N)                                                                  # :212
    pc_pred = (patches_pred.reshape(B, S, -1, 3) + rec.view(B, S, 1, 3)).reshape(B, -1, 3)         # :214-216
    d = (chamfer or chamfer_autograd)(pc_pred, x)                                                  # AE.py:67
    loss = d + lam * fbpp                                                                          # AE.py:68-70
    loss.backward()                                                                                # :229
    opt.step()                                                                                     # :230
    return dict(loss=float(loss.detach()), fbpp=float(fbpp.detach()), bpp=float(bpp.detach()))
