#!/usr/bin/env python3
"""A/B harness for the f16x2 kernels (round 3): several builds of pccx_ae_encode_h2_ws / pccx_ae_decode_h2, each in its own small
shared object, timed on the bench shape in one process and compared bit for bit with the first.

    python tools/experiments/r3/h2_variants.py build [tags]          # here (hipcc cross-compiles): tools/experiments/_build/libh2_<tag>.so
    python tools/experiments/r3/h2_variants.py run [clouds=1024] [tags]   # on the GPU box

A variant is (flags, textual patches of the product sources): a -D knob, a codegen option, or a diagnostic edit (results then differ).
"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
CSRC = os.path.join(ROOT, "point-cloud-compression_amd", "csrc")
OUT = os.path.join(ROOT, "tools", "experiments", "_build")

FAKE_SPLIT = ("h2_split8(", "h2_split8_fake(")
FAKE_DEF = r'''
__device__ __forceinline__ void h2_split8_fake(const f32x4 &v0, const f32x4 &v1, float rho, f16x8 (&pl)[2])
{   // DIAGNOSTIC: no conversion work, garbage planes (the registers still depend on the inputs)
    pl[0] = __builtin_bit_cast(f16x8, v0);
    pl[1] = __builtin_bit_cast(f16x8, v1);
}
'''
# tag -> (flags, [(file, old, new)])
VARIANTS = {
    "base": ([], []),
    "prev": ([], [("@git", "HEAD")]),              # the committed kernels (before the working-tree edits)
    "mg4": (["-DFH_MG=4"], []),
    "nb3": (["-DFH_NB=3"], []),
    "noreader": (["-DFH_READER=0"], []),
    "f2_4": (["-DFH_FIFO2=4"], []),
    "f2_12": (["-DFH_FIFO2=12"], []),
    "fifo12": (["-DFH_FIFO=12"], []),
    "fifo16": (["-DFH_FIFO=16"], []),
    "ch16": (["-DFH_CHUNK=16"], []),
    "ch48": (["-DFH_CHUNK=48"], []),
    "ch16nb4": (["-DFH_CHUNK=16", "-DFH_NB=4"], []),
    "samg4": (["-DFH_SA_MG=4"], []),
    "nosplit": ([], [("encoder_fused_h2.hip", '#define FH_CHUNK', FAKE_DEF + '#define FH_CHUNK'), ("encoder_fused_h2.hip",) + FAKE_SPLIT,
                     ("decoder_h2.hip", '#ifndef DEC_GROUP', FAKE_DEF + '#ifndef DEC_GROUP'), ("decoder_h2.hip",) + FAKE_SPLIT]),
}
_extra = os.path.join(os.path.dirname(os.path.abspath(__file__)), "h2_variants_extra.py")
if os.path.exists(_extra):
    exec(open(_extra).read())

STUB = r"""
#include <stdarg.h>
#include <stdio.h>
static char g_err[512];
void pccx_set_error(const char *fmt, ...) { va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap); }
extern "C" __attribute__((visibility("default"))) const char *encv_last_error(void) { return g_err; }
"""
FILES = ["encoder_fused_h2.hip", "decoder_h2.hip", "decoder.hip", "patch_knn.hip"]


# A diagnostic patch may not remove a PRODUCER and keep its consumers: the builds of round 3 that deleted the LDS-DMA (the weight ring was
# then never written, its reads undefined to the compiler) ended in a GPU memory fault on the box (gpurun_out/r3l/variants4.log).  Such a
# variant is refused here unless it also patches every reader of what it no longer writes.
PRODUCERS = {"__builtin_amdgcn_global_load_lds": ("ds_read_b128", "ws.chunk(", "ws.get(", "buf[")}


def check_variant(tag, patches):
    for pf, old, new in [x for x in patches if x[0] != "@git"]:
        for prod, consumers in PRODUCERS.items():
            if prod in old and prod not in new:
                touched = [c for c in consumers if any(c in o and (c not in n_ or o != n_) for _, o, n_ in [x for x in patches if x[0] != "@git"] if o is not old)]
                if len(touched) < len(consumers):
                    raise SystemExit(f"variant {tag!r} removes {prod} (the producer of the LDS weight ring) but leaves readers of the ring in place "
                                     f"({sorted(set(consumers) - set(touched))}): the ring would be read uninitialised -- refused (a round-3 build of "
                                     f"this kind faulted on the GPU box).  Separate the phases instead (onlysa / onlypn) or patch the readers too.")


def build(tags):
    os.makedirs(OUT, exist_ok=True)
    procs = []
    for t in tags:
        flags, patches = VARIANTS[t]
        check_variant(t, patches)
        d = os.path.join(OUT, "h2_" + t)
        os.makedirs(d, exist_ok=True)
        open(os.path.join(d, "stub.hip"), "w").write(STUB)
        for f in os.listdir(CSRC):
            if f.endswith(".h") or f in FILES:
                rev = [x[1] for x in patches if x[0] == "@git"]
                if rev:                                     # the file as of a git revision
                    s = subprocess.run(["git", "-C", ROOT, "show", f"{rev[0]}:point-cloud-compression_amd/csrc/{f}"], capture_output=True, text=True, check=True).stdout
                else:
                    s = open(os.path.join(CSRC, f)).read()
                for pf, old, new in [x for x in patches if x[0] != "@git"]:
                    if pf == f:
                        if old not in s:
                            raise SystemExit(f"{t}: patch of {f} does not apply: {old[:40]!r}")
                        s = s.replace(old, new) if old == FAKE_SPLIT[0] else s.replace(old, new, 1)
                s = s.replace('#include "../../include/pccx.h"', '#include "pccx.h"')
                open(os.path.join(d, f), "w").write(s)
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-honor-nans",
               "-I", os.path.join(ROOT, "include"), "-I", d, "-Wno-unused-function"] + flags + [os.path.join(d, f) for f in FILES + ["stub.hip"]] + \
              ["-o", os.path.join(OUT, f"libh2_{t}.so"), "-Rpass-analysis=kernel-resource-usage"]
        procs.append((t, subprocess.Popen(cmd, stderr=subprocess.PIPE, text=True)))
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
        print(t, "|", " || ".join(f"{k[3:30]}: " + ", ".join(v) for k, v in rows.items() if "h2_kernel" in k and "prep" not in k))


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
    enc, dec = ae._blobs(patches.device)
    eh2, dh2 = ae._enc_h2_blob(patches.device), ae._dec_h2_blob(patches.device)
    st = torch.cuda.current_stream().cuda_stream
    q = ae.encode(patches, sa_matmul="f16x2", pn_matmul="f16x2")[2]
    flop_e = 2 * K * (16 * (3 * 32 + 32 * 64 + 64 * 128) + 131 * 128 + 128 * 256 + 256 * 512 + 512 * d)
    flop_d = (d * 256 + 256 * 1024 + 1024 * k * 128) * 2 + k * (144 * 128 + 128 * 64 + 64 * 32 + 32 * 3) * 2
    ref = None
    V = C.c_void_p
    for t in tags:
        label = t
        os.environ.pop("PCCX_ENC_H2_NT", None)
        os.environ.pop("PCCX_DEC_H2_NT", None)
        while "@" in t:                          # tag@nt1: one point tile per wave in PointNet; tag@d2: two patch tiles per wave in the decoder
            t, opt = t.rsplit("@", 1)            # (forms of the same build, chosen per call by PCCX_ENC_H2_NT / PCCX_DEC_H2_NT)
            if opt == "nt1":
                os.environ["PCCX_ENC_H2_NT"] = "1"
            elif opt == "d2":
                os.environ["PCCX_DEC_H2_NT"] = "2"
        lib = C.CDLL(os.path.join(OUT, f"libh2_{t}.so"))
        lib.encv_last_error.restype = C.c_char_p
        fe = lib.pccx_ae_encode_h2_ws
        fe.argtypes = [V, C.c_int, C.c_int, V, V, C.c_int, C.c_int, V, V, V, V, V]
        lib.pccx_ae_encode_h2_workspace_bytes.restype = C.c_size_t
        lib.pccx_ae_encode_h2_workspace_bytes.argtypes = [C.c_int, C.c_int]
        wsb = torch.zeros(lib.pccx_ae_encode_h2_workspace_bytes(P, K), dtype=torch.uint8, device="cuda")
        o = [torch.zeros(P, d, device="cuda") for _ in range(3)]
        fd = lib.pccx_ae_decode_h2
        fd.argtypes = [V, C.c_int, C.c_int, C.c_int, V, V, V, V, C.c_float, V, V, V, C.c_int, C.c_double, V, V]
        lib.pccx_ae_decode_h2_workspace_floats.restype = C.c_size_t
        lib.pccx_ae_decode_h2_workspace_floats.argtypes = [C.c_int]
        wsd = torch.zeros(lib.pccx_ae_decode_h2_workspace_floats(P), device="cuda")
        od = torch.zeros(P, k, 3, device="cuda")
        calls = {"enc": lambda: fe(patches.data_ptr(), P, K, enc.data_ptr(), eh2.data_ptr(), d, L, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), wsb.data_ptr(), st),
                 "dec": lambda: fd(q.data_ptr(), P, d, k, dec.data_ptr(), dh2.data_ptr(), wsd.data_ptr(), od.data_ptr(), 0.0, None, None, None, 1, 0.01, None, st)}
        line = f"{label:14s}"
        for name, call in calls.items():
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
            best = min(ms)
            flop = flop_e if name == "enc" else flop_d
            line += f"  {name} {best:7.3f} ms ({' '.join('%.2f' % m for m in ms)}) {flop * P / best / 1e9:6.1f} TF = {flop * P / best / 1e9 / 839.0:.3f}"
        outs = [x.clone() for x in o] + [od.clone()]
        if ref is None:
            ref = outs
        same = [bool(torch.equal(x, y)) for x, y in zip(outs, ref)]
        print(line + f"  identical enc {all(same[:3])} dec {same[3]}  max|d raw| {float((outs[0] - ref[0]).abs().max()):.2e}", flush=True)


if __name__ == "__main__":
    mode = sys.argv[1] if len(sys.argv) > 1 else "build"
    if mode == "build":
        build(sys.argv[2].split(",") if len(sys.argv) > 2 else list(VARIANTS))
    else:
        clouds = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
        run(clouds, sys.argv[3].split(",") if len(sys.argv) > 3 else list(VARIANTS))
