"""CPU restatement of the per-file loop bodies of compress.py:79-155 and decompress.py:72-121.

TEST INFRASTRUCTURE ONLY -- the checker and the cpu_baseline of bench.py, never the product.

The structure of the reference is kept on purpose so that it costs what the reference costs
(BASELINE.md section 3): B = 1, a Python loop over the S patches calling ae.sa then ae.pn per
patch (compress.py:113-121), a sequential octree depth search that re-encodes at every depth
(pn_kit.py:387-395), host round trips between stages.  File reads / writes are replaced by
in-memory byte strings; PLY I/O is outside the reference's timing window anyway.
"""
import time

import numpy as np
import torch

from . import cport, ref_model


def compress_one(pc_np, ae, prob, start_idx, K=256, ALPHA=2, N0=1024, octree_mode="reference"):
    """One cloud through compress.py:82-152.  Returns (dict of streams + intermediates, seconds
    spent inside the reference's timing window :85-:154)."""
    d, L = ae.d, ae.L
    pc = torch.from_numpy(np.asarray(pc_np, dtype=np.float32)).unsqueeze(0)          # :82-83
    t0 = time.time()                                                                 # :85
    with torch.no_grad():
        pc, center, longest = ref_model.normalize(pc, margin=0.01)                   # :90
        N = pc.shape[1]
        S = int(N * ALPHA // K)                                                      # :93
        idx = ref_model.farthest_point_sample(pc, S, [start_idx])                    # :96
        sampled_xyz = ref_model.index_points(pc, idx)
        codes, _ = cport.encode_sampled_np(sampled_xyz.numpy(), 1, N, ref_model.OCTREE_BPP_DICT[K])   # :98
        rec = torch.from_numpy(cport.decode_sampled_np(codes, 1, octree_mode))       # :100-101
        assert rec.shape == sampled_xyz.shape                                        # :102
        _, knn_idx, grouped = ref_model.knn_points(rec, pc, K=K, return_nn=True)      # :70-74
        grouped = grouped - rec.view(1, S, 1, 3)
        x_patches = grouped.view(S, K, 3).transpose(1, 2)                            # :107
        x_patches = x_patches * ((N / N0) ** (1 / 3))                                # :108
        feats = []
        for j in range(S):                                                           # :113-115
            _, f = ae.sa(x_patches[j].reshape(1, 3, K))
            feats.append(f)
        feats = torch.cat(feats)
        lat = []
        for j in range(S):                                                           # :120-121
            lat.append(ae.pn(torch.cat((x_patches[j].unsqueeze(0), feats[j].unsqueeze(0)), dim=1)))
        latent_raw = torch.cat(lat)
        spread = L - 0.2                                                             # :125
        latent = torch.sigmoid(latent_raw) * spread - spread / 2
        latent_q = latent.round()                                                    # :127
        pmf = prob(rec)                                                              # :131
        cdf = ref_model.pmf_to_cdf(pmf)                                              # :134
        sym = (latent_q.view(1, S, -1).to(torch.int16) + L // 2).numpy()             # :135
        cdf_int = ref_model.cdf_float_to_int(cdf).reshape(-1, L + 1)
        p_bytes = cport.range_encode(cdf_int, sym.reshape(-1))                       # :136
        s_bytes = bytes(cport.pack_bits(codes[0]))                                   # :143-146
        c_arr = np.zeros(4)
        c_arr[:3] = center.numpy().flatten()
        c_arr[3] = float(longest)
        c_bytes = c_arr.astype(np.float32).tobytes()                                 # :149-152
    dt = time.time() - t0                                                            # :154
    return dict(s=s_bytes, p=p_bytes, c=c_bytes, fps_idx=idx[0].numpy(), bits=codes[0], rec_sampled=rec[0].numpy(),
                knn_idx=knn_idx[0].numpy(), patches=x_patches.transpose(1, 2).numpy(), latent=latent.numpy(),
                latent_q=latent_q.numpy(), cdf_int=cdf_int, pcn=pc[0].numpy()), dt


def decompress_one(s_bytes, p_bytes, c_bytes, ae, prob, N0=1024, octree_mode="reference", latent_q_override=None):
    """One cloud through decompress.py:77-118.  Returns (pc (S*k,3) f32, seconds in the window)."""
    L = ae.L
    t0 = time.time()                                                                 # :77
    with torch.no_grad():
        code = cport.unpack_bits(s_bytes)                                            # :80-82
        if octree_mode == "full":
            code = np.concatenate([code[:len(code) - 8], code[-1:]]) if len(code) else code   # undo the tail quirk
        rec = torch.from_numpy(cport.decode_sampled_np([code.astype(np.uint8)], 1, octree_mode))   # :83-84
        S = rec.shape[1]                                                             # :85
        pmf = prob(rec)                                                              # :88
        cdf_int = ref_model.cdf_float_to_int(ref_model.pmf_to_cdf(pmf)).reshape(-1, L + 1)          # :92
        if latent_q_override is None:
            sym = cport.range_decode(cdf_int, p_bytes)                               # :93
            latent = torch.from_numpy((sym.astype(np.int32) - L // 2).astype(np.float32)).view(S, -1)
        else:
            latent = torch.from_numpy(latent_q_override).view(S, -1)
        patches = ae.decode(latent)                                                  # :97-102
        k = patches.shape[1]
        N = S * k                                                                    # :106
        patches = patches / ((N / N0) ** (1 / 3))                                    # :107
        pc = (patches.view(1, S, -1, 3) + rec.view(1, S, 1, 3)).reshape(1, -1, 3)    # :110
        arr = np.frombuffer(c_bytes, dtype=np.float32)                               # :113
        center = torch.from_numpy(arr[:3].copy()).reshape(1, 3)
        longest = torch.from_numpy(arr[3:4].copy())
        pc = ref_model.denormalize(pc, center, longest, margin=0.01)                 # :116
    dt = time.time() - t0                                                            # :118
    return pc[0].numpy(), dt


def d1_psnr(orig, recon):
    """eval.py:43-98 point-to-point PSNR (the open3d KD-tree 1-NN loop as a brute-force min)."""
    d2, _ = cport.nn_dist(recon, orig)
    mse = float(np.mean(d2.astype(np.float64)))
    rng = orig.max(0).astype(np.float64) - orig.min(0).astype(np.float64)
    return 10 * np.log10(float((rng ** 2).sum()) / mse) if mse > 0 else float("inf")


def calc_uc(input_pc, decomp_pc, region=1024):
    """eval.calc_uc (eval.py:127-151): the 1024-NN region of point 0 of each cloud (pytorch3d knn_points, centred on
    the point, :130-136), each region point's distance to its nearest OTHER region point by torch.cdist + topk(k=2,
    largest=False) (:138-144), and the ratio of the population variances np.var(decomp) / np.var(input) (:146-151).
    torch.cdist keeps the reference's own call (its default compute mode forms |a|^2 + |b|^2 - 2ab for regions this size)."""
    def region_dist(pc):
        pc = torch.from_numpy(np.asarray(pc, dtype=np.float32))
        point = pc[0]
        _, _, grouped = ref_model.knn_points(point.view(1, 1, 3), pc.unsqueeze(0), K=region, return_nn=True)
        reg = (grouped - point.view(1, 1, 1, 3)).view(region, 3)
        dist = torch.cdist(reg, reg, p=2)
        values, _ = torch.topk(dist, k=2, largest=False)
        return values[:, 1].numpy()
    return float(np.var(region_dist(decomp_pc)) / np.var(region_dist(input_pc)))
