#!/usr/bin/env python3
"""decompress.py with the reference's command line (decompress.py:19-39), on the MI355X path."""
import argparse
import os
import time
from glob import glob

import numpy as np

import _common  # noqa: F401
import torch
from pccx import codec, dist, ops, plyio
from pccx._lib import PccxError

parser = argparse.ArgumentParser(prog='decompress.py', description='Deompress Point Clouds Using Trained Model.',
                                 formatter_class=argparse.ArgumentDefaultsHelpFormatter)
parser.add_argument('compressed_path', help='Comressed .bin files folder.')
parser.add_argument('decompressed_path', help='Decompressed .ply files folder.')
parser.add_argument('model_load_folder', help='Directory where to load trained models.')
_common.add_codec_flags(parser)
parser.add_argument('--S', type=int, default=None,
                    help="Patches per cloud with --octree-mode full (default: the number of centres each .s.bin holds; "
                         "'reference' mode is octree_np.decode as written, always 64).")
parser.add_argument('--bin-ply-suffix', action='store_true',
                    help="Write <name>.bin.ply (what eval.py:172 looks for) instead of <name> (decompress.py:121).")


def main():
    args = parser.parse_args()
    print(f"Processing on device (gpu/cpu): {args.device}")
    os.makedirs(args.decompressed_path, exist_ok=True)
    names = sorted(os.path.split(x)[1][:-6] for x in glob(os.path.join(args.compressed_path, '*.s.bin')))
    rank, world = _common.setup_ranks(args)
    names = [names[i] for i in dist.shard_indices(len(names), rank, world)]                      # file i -> rank i mod world
    ae, prob = _common.load_models(args)
    cd = codec.Codec(ae, prob, K=args.K, ALPHA=args.ALPHA, N0=args.N0, octree_mode=args.octree_mode)
    times = []
    with torch.no_grad():
        for b0 in range(0, len(names), args.batch):
            chunk = names[b0:b0 + args.batch]
            torch.cuda.synchronize()
            t0 = time.time()                                                                     # decompress.py:77
            # the three files of every cloud of the chunk into ONE packed host buffer by the library's host threads
            # (pccx_read_streams_host; decompress.py:80-91,113 reads them one by one), then one upload
            up = codec.Compressed.read_files(args.compressed_path, chunk, device=args.device)
            s_b, s_n, p_b, p_n, c = up.s_bytes, up.s_nbytes, up.p_bytes, up.p_nbytes, up.c
            # number of centres each stream holds (decompress.py:85 takes S from the decoded array)
            _, count = ops.octree_decode(s_b, s_n, args.octree_mode, 64 if args.octree_mode == 'reference' else 1)
            count = count.cpu().numpy()
            if (count < 0).any():
                raise PccxError("corrupt .s.bin stream(s): " + ", ".join(n for n, k_ in zip(chunk, count) if k_ < 0))
            if args.octree_mode == 'full' and (count == 0).any():
                # an empty / truncated / root-bit-0 stream decodes to NO centres: there are no patches to decode the .p.bin against
                # (S = 0 would mean zero-size launches and an empty .ply); more centres than the decoder holds come back as -1 above
                bad = [n for n, k_ in zip(chunk, count) if k_ == 0]
                raise PccxError("corrupt or empty .s.bin stream(s) (0 centres decoded; their .p.bin cannot be decoded): " + ", ".join(bad))
            S_of = np.full(len(chunk), 64) if args.octree_mode == 'reference' else (count if args.S is None else np.full(len(chunk), args.S))
            if args.octree_mode == 'full' and args.S is not None and (count != args.S).any():
                raise PccxError(f"--S {args.S} given but the streams hold {sorted(set(count.tolist()))} centres")
            clouds = [None] * len(chunk)
            for S in sorted(set(S_of.tolist())):                                                 # one launch sequence per S
                sel = torch.from_numpy(np.flatnonzero(S_of == S)).to(args.device)
                comp = codec.Compressed(s_b[sel], s_n[sel], p_b[sel], p_n[sel], c[sel], 0)
                pc = cd.decompress(comp, S=int(S)).cpu().numpy()
                for j, cloud in zip(sel.cpu().tolist(), pc):
                    clouds[j] = cloud
            times += [(time.time() - t0) / len(chunk)] * len(chunk)                              # decompress.py:118
            for n, cloud in zip(chunk, clouds):
                plyio.save_point_cloud(cloud, os.path.join(args.decompressed_path, n + ('.bin.ply' if args.bin_ply_suffix else '')))
    g = dist.gather_summaries([0.0, 0.0, 0.0, 0.0, len(times), float(np.sum(times))], _common.summary_device(args))
    if rank == 0 and float(g[:, 4].sum()) > 0:
        tot = g.sum(dim=0)
        print(f"Done! Execution time: {round(float(tot[5] / tot[4]), 5)}s per point cloud." + (f" ({int(tot[4])} clouds on {world} ranks)" if world > 1 else ""))
    _common.finish_ranks(world)


if __name__ == '__main__':
    main()
