import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
if PKG not in sys.path:
    sys.path.insert(0, PKG)

import torch  # noqa: E402


def find_checkpoint(folder, prefix):
    """The CLIs load <prefix>.pkl (compress.py:58-59) but the trainer writes <prefix>_step{N}.pkl
    (train.py:105): accept both, preferring the plain name, else the largest step."""
    p = os.path.join(folder, prefix + ".pkl")
    if os.path.exists(p):
        return p
    steps = []
    for f in os.listdir(folder) if os.path.isdir(folder) else []:
        if f.startswith(prefix + "_step") and f.endswith(".pkl"):
            try:
                steps.append((int(f[len(prefix) + 5:-4]), f))
            except ValueError:
                pass
    if not steps:
        raise FileNotFoundError(f"no {prefix}.pkl or {prefix}_step*.pkl under {folder}")
    return os.path.join(folder, max(steps)[1])


def setup_ranks(args):
    """One process per GPU: under torchrun / pccx.launch (RANK, LOCAL_RANK, WORLD_SIZE in the environment) pin this
    process to its GPU and join the process group (RCCL); files are then sharded file i -> rank i mod world
    (dist.shard_indices) and the per-rank summaries all-gathered at the end.  Returns (rank, world)."""
    from pccx import launch
    rank, local, world = launch.rank_env()
    if world > 1:
        if str(args.device).startswith("cuda"):
            local = local % max(torch.cuda.device_count(), 1)
            args.device = f"cuda:{local}"
            torch.cuda.set_device(local)
        backend = os.environ.get("PCCX_DIST_BACKEND", "nccl")
        launch.init_process_group(backend, torch.device(args.device) if backend == "nccl" else None)
    return rank, world


def finish_ranks(world):
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


def summary_device(args):
    return torch.device(args.device) if os.environ.get("PCCX_DIST_BACKEND", "nccl") == "nccl" else torch.device("cpu")


def load_models(args, need_gpu=True):
    from pccx import models
    if need_gpu and not torch.cuda.is_available():
        raise SystemExit("pccx needs a ROCm GPU: there is no CPU path")
    k = args.K // args.ALPHA
    ae = models.AE(K=args.K, k=k, d=args.d, L=args.L)
    prob = models.ConditionalProbabilityModel(args.L, args.d)
    ae.load_state_dict(torch.load(find_checkpoint(args.model_load_folder, "ae"), map_location="cpu", weights_only=True))
    prob.load_state_dict(torch.load(find_checkpoint(args.model_load_folder, "prob"), map_location="cpu", weights_only=True))
    return ae.pack(args.device), prob.pack(args.device)


def add_codec_flags(parser):
    parser.add_argument('--N0', type=int, help='Scale Transformation constant.', default=1024)
    parser.add_argument('--ALPHA', type=int, help='The factor of patch coverage ratio.', default=2)
    parser.add_argument('--K', type=int, help='Number of points in each patch.', default=256)
    parser.add_argument('--d', type=int, help='Bottleneck size.', default=16)
    parser.add_argument('--L', type=int, help='Quantization Level.', default=7)
    parser.add_argument('--device', help='AE Model Device (cuda)', default='cuda')
    parser.add_argument('--octree-mode', choices=['reference', 'full'], default='reference',
                        help="'reference' reproduces octree_np.decode as written (8 bits consumed, S=64); "
                             "'full' is the level-by-level decode.")
    parser.add_argument('--batch', type=int, default=256, help='Clouds per launch sequence.')
    parser.add_argument('--seed', type=int, default=11, help='Seed of the per-file FPS start index.')
