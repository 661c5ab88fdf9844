#!/usr/bin/env python3
"""compress.py with the reference's command line (compress.py:20-40), on the MI355X path.
Writes <name>.p.bin / .s.bin / .c.bin per input file (compress.py:139-152)."""
import argparse
import os
import time
from glob import glob

import numpy as np

import _common  # noqa: F401
import torch
from pccx import codec, dist, plyio

parser = argparse.ArgumentParser(prog='compress.py', description='Compress Point Clouds Using Trained Model.',
                                 formatter_class=argparse.ArgumentDefaultsHelpFormatter)
parser.add_argument('input_glob', help='Point clouds glob pattern for compression.')
parser.add_argument('compressed_path', help='Comressed .bin files folder.')
parser.add_argument('model_load_folder', help='Directory where to load trained models.')
_common.add_codec_flags(parser)


def main():
    args = parser.parse_args()
    print(f"Processing on device (gpu/cpu): {args.device}")
    os.makedirs(args.compressed_path, exist_ok=True)
    files = sorted(glob(args.input_glob, recursive=True))
    rank, world = _common.setup_ranks(args)
    ae, prob = _common.load_models(args)
    cd = codec.Codec(ae, prob, K=args.K, ALPHA=args.ALPHA, N0=args.N0, octree_mode=args.octree_mode)
    mine = set(dist.shard_indices(len(files), rank, world))                                      # file i -> rank i mod world
    times, todo, bits, points = [], [t for t in enumerate(files) if t[0] in mine], 0, 0
    with torch.no_grad():
        while todo:
            clouds = [(i, f, plyio.read_point_cloud(f)) for i, f in todo[:args.batch]]          # outside the timed window
            n0 = clouds[0][2].shape[0]
            batch = [c for c in clouds if c[2].shape[0] == n0]                                   # one launch = one N
            done = {c[0] for c in batch}
            todo = [t for t in todo if t[0] not in done]
            pc = torch.from_numpy(np.stack([c[2] for c in batch])).to(args.device)
            starts = [dist.fps_start_index(args.seed, c[0], n0) for c in batch]
            torch.cuda.synchronize()
            t0 = time.time()                                                                     # compress.py:85
            comp = cd.compress(pc, starts)
            # ONE D2H of the batch's packed streams, then the library's host threads cut the three files of every cloud out of it
            # (pccx_write_streams_host): <name>.p.bin / .s.bin / .c.bin, the bytes of compress.py:139-152
            nbytes = comp.write_files(args.compressed_path, [os.path.split(f)[1] for _, f, _ in batch])
            times += [(time.time() - t0) / len(batch)] * len(batch)                              # compress.py:154
            bits += 8 * nbytes
            points += n0 * len(batch)
    g = dist.gather_summaries([bits, points, 0.0, 0.0, len(times), float(np.sum(times))], _common.summary_device(args))
    if rank == 0 and float(g[:, 4].sum()) > 0:
        tot = g.sum(dim=0)
        print(f"Done! Execution time: {round(float(tot[5] / tot[4]), 5)}s per point cloud." +
              (f" ({int(tot[4])} clouds on {world} ranks, {float(tot[0] / tot[1]):.4f} bpp)" if world > 1 else ""))
    _common.finish_ranks(world)


if __name__ == '__main__':
    main()
