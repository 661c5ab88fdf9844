#!/usr/bin/env python3
"""A/B harness for the fused encoder kernel (round 3): several builds of ONE entry point, pccx_ae_encode_b3, each in its own
small shared object, timed on the bench shape in one process and compared bit for bit with the first.

    python tools/experiments/enc_variants.py build            # here (hipcc cross-compiles): tools/experiments/_build/libenc_<tag>.so
    python tools/experiments/enc_variants.py run [clouds=1024] [tags]   # on the GPU box

A variant is (source file, extra hipcc flags); sources default to the product's csrc/encoder_fused.hip, so a variant can be a
flag (-D knob or a codegen option) or a whole experimental file under tools/experiments/r3/.
"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "point-cloud-compression_amd", "csrc")
OUT = os.path.join(ROOT, "tools", "experiments", "_build")
R3 = os.path.join(ROOT, "tools", "experiments", "r3")
PRODUCT = os.path.join(CSRC, "encoder_fused.hip")

# tag -> (source, flags)
VARIANTS = {
    "base": (PRODUCT, []),
}
_extra = os.path.join(R3, "variants.py")
if os.path.exists(_extra):
    exec(open(_extra).read())          # may add to VARIANTS

STUB = r"""
#include <stdarg.h>
#include <stdio.h>
static char g_err[512];
void pccx_set_error(const char *fmt, ...) { va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap); }
extern "C" __attribute__((visibility("default"))) const char *encv_last_error(void) { return g_err; }
"""


def build(tags):
    os.makedirs(OUT, exist_ok=True)
    stub = os.path.join(OUT, "stub.hip")
    open(stub, "w").write(STUB)
    procs = []
    for t in tags:
        src, flags = VARIANTS[t]
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
               "-I", os.path.join(ROOT, "include"), "-I", CSRC, "-Wno-unused-function"] + flags + [src, stub] + ([os.path.join(CSRC, "patch_knn.hip")] if "pccx_patch_knn16" in open(src).read() else []) + ["-o", os.path.join(OUT, f"libenc_{t}.so")]
        procs.append((t, subprocess.Popen(cmd + ["-Rpass-analysis=kernel-resource-usage"], stderr=subprocess.PIPE, text=True)))
    for t, p in procs:
        err = p.communicate()[1]
        if p.returncode != 0:
            print(err[-3000:])
            raise SystemExit(f"variant {t} failed to build")
        cur, rows = None, {}
        for l in err.splitlines():
            m = l.split("remark:")[-1].replace("[-Rpass-analysis=kernel-resource-usage]", "").strip()
            if m.startswith("Function Name:"):
                cur = m.split(":")[1].strip()
                rows[cur] = []
            elif cur and any(k in m for k in ("VGPRs:", "VGPRs Spill", "ScratchSize")):
                rows[cur].append(m.replace(" [bytes/lane]", ""))
        print(t, "|", " || ".join(f"{k[-28:]}: " + ", ".join(v) for k, v in rows.items() if "sa_pn" in k or "variant" in k))


def run(clouds, tags):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "point-cloud-compression_amd"))
    import numpy as np
    import torch
    from bench import seeded_state_dict, AE_SEED, AE_LAST_GAIN
    from pccx import models, synth, ops
    K, k, d, L = 256, 128, 16, 7
    ae = models.AE(K, k, d, L)
    ae.load_state_dict(seeded_state_dict(ae, AE_SEED, last_gain=AE_LAST_GAIN))
    ae.pack("cuda")
    base = np.stack([synth.cad_cloud(11 + i, 8192) for i in range(32)])
    cl = torch.from_numpy(np.concatenate([base] * (clouds // 32 + 1))[:clouds]).cuda()
    pcn, _, _ = ops.normalize(cl)
    cent = ops.index_points(pcn, ops.farthest_point_sample_batch(pcn, 64, torch.zeros(clouds, dtype=torch.int32)))
    patches = ops.knn_points(cent, pcn, K, patch_scale=2.0).knn.view(clouds * 64, K, 3).contiguous()
    P = patches.shape[0]
    enc, sa3, pn3 = ae._blobs(patches.device)[0], ae._sa_b3_blob(patches.device), ae._pn_b3_blob(patches.device)
    st = torch.cuda.current_stream().cuda_stream
    ref = None
    flop = 2 * K * (16 * (3 * 32 + 32 * 64 + 64 * 128) + 131 * 128 + 128 * 256 + 256 * 512 + 512 * d)
    for t in tags:
        if t.endswith("!"):                      # tag! = time the workspace-free entry even when the library has the other
            os.environ["ENCV_NO_WS"] = "1"
            t = t[:-1]
        else:
            os.environ.pop("ENCV_NO_WS", None)
        lib = C.CDLL(os.path.join(OUT, f"libenc_{t}.so"))
        fn = lib.pccx_ae_encode_b3
        fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.encv_last_error.restype = C.c_char_p
        o = [torch.zeros(P, d, device="cuda") for _ in range(3)]
        call = lambda: fn(patches.data_ptr(), P, K, enc.data_ptr(), sa3.data_ptr(), pn3.data_ptr(), d, L, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), st)
        if hasattr(lib, "pccx_ae_encode_b3_ws") and not os.environ.get("ENCV_NO_WS"):     # the workspace form (neighbour tables from patch_knn.hip)
            fw = lib.pccx_ae_encode_b3_ws
            fw.argtypes = fn.argtypes[:-1] + [C.c_void_p, C.c_void_p]
            lib.pccx_ae_encode_b3_workspace_bytes.restype = C.c_size_t
            wsb = torch.zeros(lib.pccx_ae_encode_b3_workspace_bytes(P, K), dtype=torch.uint8, device="cuda")
            call = lambda: fw(patches.data_ptr(), P, K, enc.data_ptr(), sa3.data_ptr(), pn3.data_ptr(), d, L, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), wsb.data_ptr(), st)
            t = t + "+ws"
        for _ in range(2):
            rc = call()
            if rc:
                raise SystemExit(f"{t}: rc {rc}: {lib.encv_last_error().decode()}")
        torch.cuda.synchronize()
        ms = []
        for _ in range(3):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(3):
                call()
            b.record()
            torch.cuda.synchronize()
            ms.append(a.elapsed_time(b) / 3)
        if ref is None:
            ref = [x.clone() for x in o]
        same = all(torch.equal(x, y) for x, y in zip(o, ref))
        err = float((o[0] - ref[0]).abs().max())
        flips = int((o[2] != ref[2]).sum())
        best = min(ms)
        print(f"{t:24s} {best:8.3f} ms (runs {' '.join('%.2f' % m for m in ms)})  {flop * P / best / 1e9:7.1f} TFLOP/s = {flop * P / best / 1e9 / 419.5:.3f}  "
              f"bit-identical {same}  max|d raw| {err:.2e}  symbol flips {flips}", flush=True)


if __name__ == "__main__":
    mode = sys.argv[1] if len(sys.argv) > 1 else "build"
    if mode == "build":
        build(sys.argv[2].split(",") if len(sys.argv) > 2 else list(VARIANTS))
    else:
        clouds = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
        run(clouds, sys.argv[3].split(",") if len(sys.argv) > 3 else list(VARIANTS))
