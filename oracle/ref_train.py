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
    return float(loss), float(dist), float(rate)
