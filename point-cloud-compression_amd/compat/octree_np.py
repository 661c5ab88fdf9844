"""Drop-in for the reference's ``octree_np`` module (octree_np.py) on libpccx.so."""
import numpy as np
import torch

import pn_kit
from pccx import ops


def encode(pc, resolution, depth):                                # octree_np.py:10-45 at a FIXED depth
    """The bit stream of a fixed depth is the prefix-closed stream the depth search emits: run the
    kernel with min_bpp = -1 (every depth passes the rate test) is not enough (uniqueness test), so a
    fixed depth is obtained by asking for the depth-16 stream and truncating at level `depth`."""
    if resolution != 1:
        raise ValueError("pccx implements the octree for resolution=1")
    x = torch.as_tensor(np.asarray(pc, dtype=np.float32))[None].cuda()
    r = ops.octree_encode(x, 1, 1e30)                             # never accepted -> all 16 levels emitted
    bits = r["bits"][0].cpu().numpy()
    if bits[0] == 0:
        return bits[:1].copy()
    pos, occ = 1, 1
    for _ in range(depth):
        n = 8 * occ
        occ = int(bits[pos:pos + n].sum())
        pos += n
    return bits[:pos].copy()


def decode(bits, resolution):                                     # octree_np.py:47-112 (as written)
    return pn_kit.decode_sampled_np([np.asarray(bits, dtype=np.uint8)], resolution, "reference")[0]


def getDecodeFromPc(pc, resolution, depth):                       # octree_np.py:114-133
    pc = np.asarray(pc, dtype=np.float32)
    cube = np.float32(resolution / max(1.0, 2.0 ** min(depth, 30)))
    return np.unique(np.nan_to_num(pc // cube * cube + cube / 2), axis=0)
