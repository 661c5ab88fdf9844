#!/usr/bin/env python3
"""decompress.py with the reference's command line (decompress.py:19-39), on the MI355X path."""
import argparse
import os
import time
from glob import glob

import numpy as np

import _common  # noqa: F401
import torch
from pccx import codec, plyio

parser = argparse.ArgumentParser(prog='decompress.py', description='Deompress Point Clouds Using Trained Model.',
                                 formatter_class=argparse.ArgumentDefaultsHelpFormatter)
parser.add_argument('compressed_path', help='Comressed .bin files folder.')
parser.add_argument('decompressed_path', help='Decompressed .ply files folder.')
parser.add_argument('model_load_folder', help='Directory where to load trained models.')
_common.add_codec_flags(parser)
parser.add_argument('--S', type=int, default=64, help="Patches per cloud (only used with --octree-mode full).")
parser.add_argument('--bin-ply-suffix', action='store_true',
                    help="Write <name>.bin.ply (what eval.py:172 looks for) instead of <name> (decompress.py:121).")


def _pad(rows, dev):
    n = max(len(r) for r in rows)
    out = np.zeros((len(rows), max(n, 1)), dtype=np.uint8)
    for i, r in enumerate(rows):
        out[i, :len(r)] = np.frombuffer(r, dtype=np.uint8)
    return torch.from_numpy(out).to(dev), torch.tensor([len(r) for r in rows], dtype=torch.int32, device=dev)


def main():
    args = parser.parse_args()
    print(f"Processing on device (gpu/cpu): {args.device}")
    os.makedirs(args.decompressed_path, exist_ok=True)
    names = sorted(os.path.split(x)[1][:-6] for x in glob(os.path.join(args.compressed_path, '*.s.bin')))
    ae, prob = _common.load_models(args)
    cd = codec.Codec(ae, prob, K=args.K, ALPHA=args.ALPHA, N0=args.N0, octree_mode=args.octree_mode)
    times = []
    with torch.no_grad():
        for b0 in range(0, len(names), args.batch):
            chunk = names[b0:b0 + args.batch]
            torch.cuda.synchronize()
            t0 = time.time()                                                                     # decompress.py:77
            rd = lambda n, e: open(os.path.join(args.compressed_path, n + e), 'rb').read()
            s_b, s_n = _pad([rd(n, '.s.bin') for n in chunk], args.device)
            p_b, p_n = _pad([rd(n, '.p.bin') for n in chunk], args.device)
            c = torch.from_numpy(np.stack([np.frombuffer(rd(n, '.c.bin'), dtype=np.float32) for n in chunk])).to(args.device)
            comp = codec.Compressed(s_b, s_n, p_b, p_n, c, 0)
            pc = cd.decompress(comp, S=64 if args.octree_mode == 'reference' else args.S).cpu().numpy()
            times += [(time.time() - t0) / len(chunk)] * len(chunk)                              # decompress.py:118
            for n, cloud in zip(chunk, pc):
                plyio.save_point_cloud(cloud, os.path.join(args.decompressed_path, n + ('.bin.ply' if args.bin_ply_suffix else '')))
    if times:
        print(f"Done! Execution time: {round(float(np.mean(times)), 5)}s per point cloud.")


if __name__ == '__main__':
    main()
