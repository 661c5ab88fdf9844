#!/usr/bin/env python3
"""train.py with the reference's command line (train.py:22-50), on the MI355X path: ``--model AE`` (the IPDAE autoencoder) trains through
pccx.train_ipdae; checkpoints carry the reference's names (ae_step{N}.pkl, prob_step{N}.pkl, optimizer_step{N}.pkl, global_step{N}.pkl,
train.py:103-108) and the reference's state_dict keys, so compress.py / decompress.py load them unchanged."""
import argparse
import os
from glob import glob

import numpy as np

import _common  # noqa: F401
import torch
from pccx import models, plyio, train_ipdae

torch.manual_seed(11)                                                                  # train.py:17-19
np.random.seed(11)

parser = argparse.ArgumentParser(prog='train_ae.py', description='Train autoencoder using point cloud patches',
                                 formatter_class=argparse.ArgumentDefaultsHelpFormatter)
parser.add_argument('--train_glob', default='./data/ModelNet40_pc_01_8192p/**/train/*.ply', help='Point clouds glob pattern for training.')
parser.add_argument('--model_save_folder', default='./model/K256/', help='Directory where to save trained models.')
parser.add_argument('--model', default='AE', help='Type of the model (AE or PPPF-AE).')
parser.add_argument('--N', type=int, default=8192, help='Point cloud resolution.')
parser.add_argument('--N0', type=int, default=1024, help='Scale Transformation constant.')
parser.add_argument('--ALPHA', type=int, default=2, help='The factor of patch coverage ratio.')
parser.add_argument('--K', type=int, default=256, help='Number of points in each patch.')
parser.add_argument('--d', type=int, default=16, help='Bottleneck size.')
parser.add_argument('--L', type=int, default=7, help='Quantization Level.')
parser.add_argument('--lr', type=float, default=0.0005, help='Learning rate.')
parser.add_argument('--batch_size', type=int, default=1, help='Batch size (must be 1).')
parser.add_argument('--step_window', type=float, default=100, help='Number of steps per window to iterate in epoch.')
parser.add_argument('--lamda', type=float, default=1e-06, help='Lambda for rate-distortion tradeoff.')
parser.add_argument('--rate_loss_enable_step', type=int, default=40000, help='Apply rate-distortion tradeoff at x steps.')
parser.add_argument('--lr_decay', type=float, default=0.1, help='Decays the learning rate to x times the original.')
parser.add_argument('--lr_decay_steps', type=int, default=60000, help='Decays the learning rate every x steps.')
parser.add_argument('--max_steps', type=int, default=80000, help='Train up to this number of steps.')
parser.add_argument('--device', default='cuda', help='AE Model Device (cuda)')
parser.add_argument('--reset', action='store_true', help='Reset training and start from scratch (ignore saved model).')
parser.add_argument('--autocast', action='store_true', help='bf16 operands on the matrix cores (the reference wraps its CUDA step in autocast, train.py:175).')
parser.add_argument('--eager', action='store_true', help='launch every kernel of every step from Python instead of replaying the captured step.')


def latest(folder, prefix):
    """train.py:67-77: the file <prefix>_step{largest N}.pkl, or '' when there is none"""
    steps = []
    for f in os.listdir(folder):
        if f.startswith(prefix + "_step") and f.endswith(".pkl"):
            try:
                steps.append(int(f[len(prefix) + 5:-4]))
            except ValueError:
                pass                                                                   # the final dump's "<prefix>_step.pkl"
    return os.path.join(folder, f"{prefix}_step{max(steps)}.pkl") if steps else ''


def load_checkpoints(tr, folder):                                                      # train.py:80-100
    start = 0
    for prefix, target in (("ae", tr.ae), ("prob", tr.prob), ("optimizer", tr.opt)):
        p = latest(folder, prefix)
        if p:
            print(f"Loading {prefix} from:", p)
            sd = torch.load(p, map_location="cpu", weights_only=True)                  # tensors, numbers and containers only
            target.load_state_dict(sd)
    p = latest(folder, "global")
    if p:
        start = int(torch.load(p, map_location="cpu", weights_only=True)) + 1          # :97-99
        print("Starting step:", start)
    return start


def dump_checkpoints(tr, folder, global_step=''):                                      # train.py:103-108
    torch.save(tr.ae.state_dict(), os.path.join(folder, f'ae_step{global_step}.pkl'))
    torch.save(tr.prob.state_dict(), os.path.join(folder, f'prob_step{global_step}.pkl'))
    torch.save(tr.opt.state_dict(), os.path.join(folder, f'optimizer_step{global_step}.pkl'))
    torch.save(global_step, os.path.join(folder, f'global_step{global_step}.pkl'))


def main():
    args = parser.parse_args()
    if args.model != 'AE':
        raise SystemExit(f"--model {args.model}: only AE (the IPDAE autoencoder, AE.py) trains on this path; PPPF-AE's train-mode BatchNorm "
                         f"stacks are not built (pccx/train_ipdae.py)")
    if not torch.cuda.is_available():
        raise SystemExit("pccx needs a ROCm GPU: there is no CPU path")
    N, K = args.N, args.K
    args.S, args.k = N * args.ALPHA // K, K // args.ALPHA                              # train.py:253
    print(f"Training {args.model} on {args.device}")
    print(f"N={N}, K={K}, S={args.S}, d={args.d}, L={args.L}")
    os.makedirs(args.model_save_folder, exist_ok=True)
    files = sorted(glob(args.train_glob, recursive=True))
    points = np.stack([plyio.read_point_cloud(f) for f in files]).astype(np.float32)   # pn_kit.read_point_clouds (pn_kit.py:36-42)
    print(f"Loaded {points.shape} points, range: [{points.min()}, {points.max()}]")
    data = torch.from_numpy(points).to(args.device)                                    # the whole training set lives in HBM
    ae = models.AE(K=K, k=args.k, d=args.d, L=args.L).to(args.device)
    prob = models.ConditionalProbabilityModel(args.L, args.d).to(args.device)
    tr = train_ipdae.IpdaeTrainer(ae, prob, N=N, N0=args.N0, ALPHA=args.ALPHA, K=K, lr=args.lr, lamda=args.lamda,
                                  rate_loss_enable_step=args.rate_loss_enable_step, lr_decay=args.lr_decay,
                                  lr_decay_steps=args.lr_decay_steps, autocast=args.autocast)
    if not args.reset:
        tr.global_step = load_checkpoints(tr, args.model_save_folder)                  # :137-144
        print(f"Resuming from step {tr.global_step}")
    else:
        print("Resetting training from scratch.")
    losses, fbpps, bpps = [], [], []
    graph = None                       # the iteration as one hipGraph (pccx.train_ipdae.GraphedIpdaeStep), captured on the first full batch
    for epoch in range(9999):
        order = torch.randperm(len(files))                                             # DataLoader(shuffle=True), train.py:117
        for b0 in range(0, len(files), args.batch_size):
            if tr.global_step > args.max_steps:                                        # :163-164
                break
            batch = data[order[b0:b0 + args.batch_size].to(args.device)]
            lr_before = tr.lr
            starts = torch.randint(0, N, (batch.shape[0],), dtype=torch.long)         # the draw of pn_kit.py:321
            if args.eager or batch.shape[0] != args.batch_size:
                out = tr.step(batch, starts)                                           # (a short last batch has another shape: eager)
            elif graph is None:
                graph = tr.graphed(batch, starts, warmup=1)                            # the warm-up iteration IS this step
                out = {k_: float(v) for k_, v in zip(("loss", "fbpp", "bpp"), graph.warm_out)}
            else:
                out = graph(batch, starts)
            losses.append(out["loss"]), fbpps.append(out["fbpp"]), bpps.append(out["bpp"])
            if tr.global_step % args.step_window == 0:                                 # :241-247
                print(f"[Epoch {epoch}] Step {tr.global_step} | Feature bpp: {np.mean(fbpps):.5f} | Bpp: {np.mean(bpps):.5f} | "
                      f"Loss: {np.mean(losses):.5f}")
                losses, fbpps, bpps = [], [], []
                dump_checkpoints(tr, args.model_save_folder, tr.global_step)
            if tr.lr != lr_before:
                print(f"LR decayed to {tr.lr} at step {tr.global_step}")               # :254
        if tr.global_step > args.max_steps:
            break
    dump_checkpoints(tr, args.model_save_folder)                                       # :286 (the final, un-numbered dump)


if __name__ == '__main__':
    main()
