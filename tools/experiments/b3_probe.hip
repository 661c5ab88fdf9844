// b3_probe.hip -- experiment (not part of libpccx): fp32 products on the bf16 matrix cores.
//   1. checks the operand layout assumed for v_mfma_f32_16x16x32_bf16 (A: row = lane%16, k = 8*(lane/16)+j;
//      B: col = lane%16, same k; C/D: col = lane%16, rows 4*(lane/16)+r) against a host product;
//   2. splits fp32 operands into three bf16 pieces (x = hi + mid + lo exactly) and accumulates the six products with
//      i + j <= 4 in fp32, and compares the error against float64 with that of v_mfma_f32_16x16x4_f32 on the same data;
//   3. times both forms on a K=1024 dot-product chain (operands in registers, 4 accumulators, 1 wave per SIMD x 1024).
// Build/run:  hipcc --offload-arch=gfx950 -O3 tools/experiments/b3_probe.hip -o /tmp/b3_probe && /tmp/b3_probe
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __host__ inline unsigned short bf16_rne(float x)
{
    unsigned u;
    memcpy(&u, &x, 4);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ __host__ inline float bf16_to_f32(unsigned short h)
{
    unsigned u = (unsigned)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
__device__ __host__ inline void split3(float x, unsigned short &h, unsigned short &m, unsigned short &l)
{
    h = bf16_rne(x);
    const float r1 = x - bf16_to_f32(h);
    m = bf16_rne(r1);
    const float r2 = r1 - bf16_to_f32(m);
    l = bf16_rne(r2);
}

// C[16x16] = A[16xK] * B[Kx16], K multiple of 32; A row-major [16][K], B [K][16]
__global__ void gemm_b3(const float *A, const float *B, int K, float *C, int passes)
{
    const int lane = threadIdx.x, i = lane & 15, kg = lane >> 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < K; k0 += 32) {
        unsigned short ah[8], am[8], al[8], bh[8], bm[8], bl[8];
        for (int j = 0; j < 8; ++j) {
            split3(A[i * K + k0 + 8 * kg + j], ah[j], am[j], al[j]);
            split3(B[(k0 + 8 * kg + j) * 16 + i], bh[j], bm[j], bl[j]);
        }
        bf16x8 Ah, Am, Al, Bh, Bm, Bl;
        memcpy(&Ah, ah, 16); memcpy(&Am, am, 16); memcpy(&Al, al, 16);
        memcpy(&Bh, bh, 16); memcpy(&Bm, bm, 16); memcpy(&Bl, bl, 16);
        // smallest terms first
        if (passes >= 6) {
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Al, Bh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ah, Bl, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Am, Bm, acc, 0, 0, 0);
        }
        if (passes >= 3) {
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Am, Bh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ah, Bm, acc, 0, 0, 0);
        }
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ah, Bh, acc, 0, 0, 0);
    }
    for (int r = 0; r < 4; ++r) C[(4 * kg + r) * 16 + i] = acc[r];      // row 4*kg + r, column i
}

__global__ void gemm_f32(const float *A, const float *B, int K, float *C)
{
    const int lane = threadIdx.x, i = lane & 15, kg = lane >> 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < K; k0 += 4)
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[i * K + k0 + kg], B[(k0 + kg) * 16 + i], acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) C[(4 * kg + r) * 16 + i] = acc[r];
}

// throughput: 1024 workgroups x 4 waves, each wave runs `iters` k-steps on 8 independent accumulators
__global__ void rate_b3(int iters, float *out)
{
    bf16x8 a[3], b[3];
    for (int p = 0; p < 3; ++p)
        for (int j = 0; j < 8; ++j) { a[p][j] = (__bf16)(0.001f * (threadIdx.x + j + p)); b[p][j] = (__bf16)(0.002f * (threadIdx.x + 2 * j + p)); }
    f32x4 acc[8];
    for (int t = 0; t < 8; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], b[0], acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[2], acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[1], acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[0], acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[1], acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[0], acc[t], 0, 0, 0);
        }
    float s = 0.f;
    for (int t = 0; t < 8; ++t) s += acc[t][0];
    if (s == 12345.f) out[0] = s;
}
__global__ void rate_f32(int iters, float *out)
{
    float a[8], b[8];
    for (int j = 0; j < 8; ++j) { a[j] = 0.001f * (threadIdx.x + j); b[j] = 0.002f * (threadIdx.x + 2 * j); }
    f32x4 acc[8];
    for (int t = 0; t < 8; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j], acc[t], 0, 0, 0);   // K = 32 per (it, t)
    float s = 0.f;
    for (int t = 0; t < 8; ++t) s += acc[t][0];
    if (s == 12345.f) out[0] = s;
}

int main()
{
    const int K = 1024;
    std::vector<float> A(16 * K), B(K * 16), C(256);
    srand(1);
    for (auto &v : A) v = (float)rand() / RAND_MAX * 2.f - 1.f;
    for (auto &v : B) v = ((float)rand() / RAND_MAX * 2.f - 1.f) * (rand() % 7 == 0 ? 1e-3f : 1.f);
    float *dA, *dB, *dC;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, 1024);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    std::vector<double> ref(256, 0.0), mag(256, 0.0);
    for (int i = 0; i < 16; ++i)
        for (int n = 0; n < 16; ++n)
            for (int k = 0; k < K; ++k) { ref[i * 16 + n] += (double)A[i * K + k] * B[k * 16 + n]; mag[i * 16 + n] += fabs((double)A[i * K + k] * B[k * 16 + n]); }
    auto report = [&](const char *name) {
        hipMemcpy(C.data(), dC, 1024, hipMemcpyDeviceToHost);
        double e = 0, er = 0;
        for (int t = 0; t < 256; ++t) { e = fmax(e, fabs(C[t] - ref[t])); er = fmax(er, fabs(C[t] - ref[t]) / mag[t]); }
        printf("%-28s max |err| %.3e   max |err| / sum|a*b| %.3e\n", name, e, er);
    };
    hipLaunchKernelGGL(gemm_f32, dim3(1), dim3(64), 0, 0, dA, dB, K, dC); report("fp32 MFMA 16x16x4");
    hipLaunchKernelGGL(gemm_b3, dim3(1), dim3(64), 0, 0, dA, dB, K, dC, 6); report("bf16x3, 6 products");
    hipLaunchKernelGGL(gemm_b3, dim3(1), dim3(64), 0, 0, dA, dB, K, dC, 3); report("bf16x2-ish, 3 products");
    hipLaunchKernelGGL(gemm_b3, dim3(1), dim3(64), 0, 0, dA, dB, K, dC, 1); report("bf16, 1 product");

    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    for (int form = 0; form < 2; ++form) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (form == 0) hipLaunchKernelGGL(rate_f32, dim3(2048), dim3(256), 0, 0, iters, dC);
            else hipLaunchKernelGGL(rate_b3, dim3(2048), dim3(256), 0, 0, iters, dC);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
        }
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double flop = 2048.0 * 4 * iters * 8 * (2.0 * 16 * 16 * 32);       // fp32-equivalent FLOP (K = 32 per step)
        printf("%-28s %.3f ms  %.1f TFLOP/s fp32-equivalent%s\n", form ? "bf16x3 six-pass" : "fp32 MFMA", ms, flop / ms / 1e9,
               form ? "  (x6 = bf16 MFMA rate)" : "");
    }
    return 0;
}
