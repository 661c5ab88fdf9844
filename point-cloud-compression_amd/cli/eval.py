#!/usr/bin/env python3
"""eval.py with the reference's command line (eval.py:20-35): D1 (point-to-point) PSNR, bpp from the
three file sizes, D2 (point-to-plane) PSNR with 30-NN PCA normals, Chamfer distance on min-max-normalised
clouds, uniformity coefficient -> CSV with the reference's columns."""
import argparse
import os
from glob import glob

import numpy as np
import pandas as pd

import _common  # noqa: F401
import torch
from pccx import codec, dist, ops, plyio

parser = argparse.ArgumentParser(prog='eval.py', description='Evaluate point cloud patches',
                                 formatter_class=argparse.ArgumentDefaultsHelpFormatter)
parser.add_argument('--input_glob', default='./data/ModelNet40_pc_01_8192p/**/test/*.ply')
parser.add_argument('--compressed_path', default='./data/ModelNet40_K256_compressed/')
parser.add_argument('--decompressed_path', default='./data/ModelNet40_K256_decompressed/')
parser.add_argument('--output_file', default='./eval/ModelNet40_K256.csv')
parser.add_argument('--device', default='cuda')


def calc_uc(input_pc, decomp_pc):
    """eval.py:127-151: variance ratio of nearest-neighbour distances inside the 1024-NN region of point 0."""
    def nn_var(pc):
        K = min(1024, pc.shape[1])
        region = ops.knn_points(pc[:, :1].contiguous(), pc, K, patch_scale=1.0).knn[:, 0]   # (1,K,3), centred on point 0 (:130-136)
        d2 = ops.knn_points(region, region, 2).dists[..., 1]                        # nearest other point (:138-144)
        return torch.sqrt(d2).double().var(unbiased=False)
    return float(nn_var(decomp_pc) / nn_var(input_pc))


def main():
    args = parser.parse_args()
    print(f"Processing on device (gpu/cpu): {args.device}")
    files = sorted(glob(args.input_glob, recursive=True))
    rank, world = _common.setup_ranks(args)
    rows = []
    for f in [files[i] for i in dist.shard_indices(len(files), rank, world)]:                 # file i -> rank i mod world
        name = os.path.split(f)[1]
        cand = [os.path.join(args.decompressed_path, name + '.bin.ply'), os.path.join(args.decompressed_path, name)]
        decomp_f = next((c for c in cand if os.path.exists(c)), None)                # eval.py:172 vs decompress.py:121
        if decomp_f is None:
            continue
        a = torch.from_numpy(plyio.read_point_cloud(f))[None].to(args.device)
        b = torch.from_numpy(plyio.read_point_cloud(decomp_f))[None].to(args.device)
        bits = sum(os.stat(os.path.join(args.compressed_path, name + e)).st_size * 8 for e in ('.s.bin', '.p.bin', '.c.bin'))
        rows.append(dict(filename=name, p2pointPSNR=round(float(codec.d1_psnr(a, b)[0]), 3), p2planePSNR=round(float(codec.d2_psnr(a, b)[0]), 3),
                         chamfer_distance=float(codec.normalized_chamfer(a, b)[0]), n_points_input=a.shape[1],
                         n_points_output=b.shape[1], bpp=bits / a.shape[1],                 # eval.py:189
                         **{'uniformity coefficient': round(calc_uc(a, b), 3)}))
    if world > 1:                                            # per-file rows travel to rank 0 (a few hundred bytes per file)
        import torch.distributed as tdist
        gathered = [None] * world
        tdist.all_gather_object(gathered, rows)
        rows = sorted((r for part in gathered for r in part), key=lambda r: r["filename"])
        _common.finish_ranks(world)
        if rank != 0:
            return
    df = pd.DataFrame(rows, columns=['filename', 'p2pointPSNR', 'p2planePSNR', 'chamfer_distance', 'n_points_input',
                                     'n_points_output', 'bpp', 'uniformity coefficient'])
    if len(df):
        print(f"Done! The average p2pointPSNR: {round(df.p2pointPSNR.mean(), 3)} | p2plane PSNR: {round(df.p2planePSNR.mean(), 3)} | chamfer distance: "
              f"{round(df.chamfer_distance.mean(), 8)} | bpp: {round(df.bpp.mean(), 3)} | uc: {round(df['uniformity coefficient'].mean(), 3)}")
    os.makedirs(os.path.dirname(os.path.abspath(args.output_file)), exist_ok=True)
    df.to_csv(args.output_file)
    print(f"Evaluation results saved to {args.output_file}")


if __name__ == '__main__':
    main()
