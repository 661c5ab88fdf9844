"""Builds the DIAGNOSTIC libraries libpccx_stamps0.so / libpccx_stamps256.so beside libpccx.so: the fused encoder with s_memtime stamps at
its phase boundaries and around the SetAbstraction units, accumulated by one observed thread (0 = wave 0, 256 = wave 4: the two waves of
SIMD 0) into a __device__ array read by fused_stamps.py.  The product kernel carries no stamps.  Run from the repo root after
`python -m pccx.build`; then `gpurun -- 'bash tools/experiments/run_stamps.sh'`."""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "point-cloud-compression_amd"))
from pccx import build as b  # noqa: E402

s = open(os.path.join(b.CSRC, "encoder_fused.hip")).read()


def rep(old, new):
    global s
    assert old in s, old[:70]
    s = s.replace(old, new, 1)


rep('''__global__ __launch_bounds__(512, 1) void sa_pn_forward_b3_kernel(''', '''__device__ unsigned long long fu_stamps[16];
extern "C" __attribute__((visibility("default"))) int pccx_debug_fused_stamps(unsigned long long *out16, int reset)
{
    if (out16 && hipMemcpyFromSymbol(out16, HIP_SYMBOL(fu_stamps), sizeof(unsigned long long) * 16) != hipSuccess) return -1;
    if (reset) { unsigned long long z[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(fu_stamps), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
#ifndef FU_WHO
#define FU_WHO 0
#endif
#define FU_SUB(k) do { const unsigned long long t2_ = __builtin_amdgcn_s_memtime(); if (tid == FU_WHO) atomicAdd(&fu_stamps[k], t2_ - t_sub); t_sub = t2_; } while (0)
#define FU_STAMP(k) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); if (tid == 0) atomicAdd(&fu_stamps[k], t_ - t_prev); t_prev = t_; } while (0)
__global__ __launch_bounds__(512, 1) void sa_pn_forward_b3_kernel(''')
rep('''    for (int i = tid; i < 3 * K; i += 512) sx[i] = xp[i];
    if (tid < 8) sa_next[tid] = 0;
    __syncthreads();
''', '''    unsigned long long t_prev = __builtin_amdgcn_s_memtime();
    for (int i = tid; i < 3 * K; i += 512) sx[i] = xp[i];
    if (tid < 8) sa_next[tid] = 0;
    __syncthreads();
    FU_STAMP(0);
''')
rep('''    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    if (lane < 16) smax[wu][lane] = -INFINITY;''', '''    FU_STAMP(1);
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    if (lane < 16) smax[wu][lane] = -INFINITY;''')
rep('''        for (;;) {
            int unit = 0;''', '''        unsigned long long t_sub = __builtin_amdgcn_s_memtime();
        for (;;) {
            int unit = 0;''')
rep('''            if (unit >= units) break;
''', '''            if (unit >= units) break;
            if (tid == FU_WHO) atomicAdd(&fu_stamps[6], 1ull);
''')
rep('''        __syncthreads();                                  // every row of the pass is staged
''', '''        __builtin_amdgcn_sched_barrier(0); FU_SUB(8); __builtin_amdgcn_sched_barrier(0);
        __syncthreads();                                  // every row of the pass is staged
        FU_SUB(9);
''')
rep('''        __syncthreads();                                  // every wave has its tile in registers: the region becomes the weight ring
''', '''        __builtin_amdgcn_sched_barrier(0); FU_SUB(10); __builtin_amdgcn_sched_barrier(0);
        __syncthreads();                                  // every wave has its tile in registers: the region becomes the weight ring
        FU_SUB(11);
        FU_STAMP(2);
''')
rep('''        __syncthreads();                                  // every wave is done reading the ring: the region is staging again
    }''', '''        __syncthreads();                                  // every wave is done reading the ring: the region is staging again
        FU_STAMP(3);
    }''')
rep('''    __syncthreads();                                      // smax / sx / nbr16 are rewritten for the next patch
  }''', '''    __syncthreads();                                      // smax / sx / nbr16 are rewritten for the next patch
    FU_STAMP(4);
    if (tid == 0) atomicAdd(&fu_stamps[7], 1ull);
  }''')
tmp = "/tmp/pccx_stamped"
os.makedirs(tmp, exist_ok=True)
for h in os.listdir(b.CSRC):
    if h.endswith(".h"):
        shutil.copy(os.path.join(b.CSRC, h), os.path.join(tmp, h))
txt = open(os.path.join(tmp, "common.h")).read().replace('#include "../../include/pccx.h"', '#include "pccx.h"')
open(os.path.join(tmp, "common.h"), "w").write(txt)
open(os.path.join(tmp, "encoder_fused.hip"), "w").write(s)
for who in (0, 256):
    obj = os.path.join(tmp, f"ef_{who}.o")
    subprocess.check_call([b.HIPCC] + b.FLAGS + [f"-DFU_WHO={who}", "-c", os.path.join(tmp, "encoder_fused.hip"), "-o", obj])
    objs = [os.path.join(b.OBJ, f) for f in sorted(os.listdir(b.OBJ)) if f.endswith(".o") and not f.startswith("encoder_fused")] + [obj]
    subprocess.check_call([b.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(b.LIBDIR, f"libpccx_stamps{who}.so")] + objs)
    print("built", f"libpccx_stamps{who}.so")
