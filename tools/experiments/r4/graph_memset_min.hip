// Minimal reproducer for the round-3 hipGraph hazard, independent of torch and of libpccx (round 4).
//   chain captured from ONE stream, LINKS links (-DLINKS=256 -DROWS=512 -DBIG for a graph of the training step's size, with a 4 MB memset per link too):   hipMemsetAsync(sums) -> accumulate(sums += x_l) -> finalize(out_l = sums)
// `sums` is one buffer reused by every link, as the training step's column-sum scratch is.  The program prints the graph's nodes and
// edges (hipGraphGetNodes / hipGraphGetEdges: is every memset node ordered after the previous finalize and before its accumulate?),
// then replays the graph back to back and after idle gaps (a host sleep + a small unrelated kernel on the stream) and checks every
// out_l against the host sum.  Run it with and without DEBUG_CLR_GRAPH_PACKET_CAPTURE=0; with -DZERO_KERNEL the clear is a kernel node.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <unistd.h>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

#ifndef LINKS
#define LINKS 48
#endif
#ifndef ROWS
#define ROWS 4096
#endif
constexpr int C = 256;

__global__ void accumulate(float *sums, const float *x, int rows)
{
    const int c = threadIdx.x;
    float s = 0.f;
    for (int r = blockIdx.x; r < rows; r += gridDim.x) s += x[(size_t)r * C + c];
    atomicAdd(&sums[c], s);
}
__global__ void finalize(const float *sums, float *out) { out[threadIdx.x] = sums[threadIdx.x]; }
__global__ void zero_kernel(float *p) { p[threadIdx.x] = 0.f; }
__global__ void touch(float *p) { p[threadIdx.x] += 1.f; }

int main()
{
    hipStream_t st;
    CHECK(hipStreamCreate(&st));
    float *x, *sums, *out, *other;
    CHECK(hipMalloc(&x, (size_t)LINKS * ROWS * C * 4));
    CHECK(hipMalloc(&sums, C * 4));
    CHECK(hipMalloc(&out, (size_t)LINKS * C * 4));
    CHECK(hipMalloc(&other, C * 4));
    float *big;
    CHECK(hipMalloc(&big, (size_t)4 << 20));
    std::vector<float> hx((size_t)LINKS * ROWS * C);
    unsigned s = 12345u;
    for (auto &v : hx) { s = s * 1664525u + 1013904223u; v = (float)((s >> 20) & 7); }          // small integers: every partial sum is exact
    CHECK(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
    std::vector<float> want((size_t)LINKS * C, 0.f);
    for (int l = 0; l < LINKS; ++l)
        for (int r = 0; r < ROWS; ++r)
            for (int c = 0; c < C; ++c) want[(size_t)l * C + c] += hx[((size_t)l * ROWS + r) * C + c];

    hipGraph_t g;
    CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int l = 0; l < LINKS; ++l) {
#ifdef ZERO_KERNEL
        hipLaunchKernelGGL(zero_kernel, dim3(1), dim3(C), 0, st, sums);
#else
        CHECK(hipMemsetAsync(sums, 0, C * 4, st));
#endif
#ifdef BIG
        CHECK(hipMemsetAsync(big, 0, (size_t)4 << 20, st));          // a large clear, as the scatter-add outputs of the step have
        hipLaunchKernelGGL(touch, dim3(1), dim3(C), 0, st, big + (l % 1000) * C);
#endif
        hipLaunchKernelGGL(accumulate, dim3(64), dim3(C), 0, st, sums, x + (size_t)l * ROWS * C, ROWS);
        hipLaunchKernelGGL(finalize, dim3(1), dim3(C), 0, st, sums, out + (size_t)l * C);
    }
    CHECK(hipStreamEndCapture(st, &g));
    size_t nn = 0, ne = 0;
    CHECK(hipGraphGetNodes(g, nullptr, &nn));
    std::vector<hipGraphNode_t> nodes(nn);
    CHECK(hipGraphGetNodes(g, nodes.data(), &nn));
    CHECK(hipGraphGetEdges(g, nullptr, nullptr, &ne));
    std::vector<hipGraphNode_t> from(ne), to(ne);
    CHECK(hipGraphGetEdges(g, from.data(), to.data(), &ne));
    int nk = 0, nm = 0, bad = 0;
    for (auto n : nodes) {
        hipGraphNodeType t;
        CHECK(hipGraphNodeGetType(n, &t));
        int in = 0, outd = 0;
        for (size_t e = 0; e < ne; ++e) { in += to[e] == n; outd += from[e] == n; }
        if (t == hipGraphNodeTypeMemset) { ++nm; bad += (in != 1 && n != nodes[0]) || outd != 1; }
        else if (t == hipGraphNodeTypeKernel) ++nk;
    }
    printf("graph: %zu nodes (%d kernel, %d memset), %zu edges (a chain has nodes - 1 = %zu); memset nodes without exactly one successor / predecessor: %d\n",
           nn, nk, nm, ne, nn - 1, bad);
    hipGraphExec_t ex;
    CHECK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
    std::vector<float> got((size_t)LINKS * C);
    auto check = [&](const char *what, int rep) {
        CHECK(hipStreamSynchronize(st));
        CHECK(hipMemcpy(got.data(), out, got.size() * 4, hipMemcpyDeviceToHost));
        int wrong = 0, first = -1;
        for (size_t i = 0; i < got.size(); ++i)
            if (got[i] != want[i]) { if (first < 0) first = (int)(i / C); ++wrong; }
        if (wrong) printf("  %s replay %d: %d of %zu outputs wrong, first wrong link %d (got %.1f want %.1f)\n", what, rep, wrong, got.size(), first,
                          got[(size_t)first * C], want[(size_t)first * C]);
        CHECK(hipMemset(out, 0xff, got.size() * 4));
        return wrong;
    };
    int wb = 0, wg = 0;
    for (int rep = 0; rep < 8; ++rep) {                   // back to back: four launches, one check
        for (int q = 0; q < 4; ++q) CHECK(hipGraphLaunch(ex, st));
        wb += check("back-to-back", rep) != 0;
    }
    for (int rep = 0; rep < 24; ++rep) {                  // after an idle gap and an unrelated kernel on the stream
        usleep(20000);
        hipLaunchKernelGGL(touch, dim3(1), dim3(C), 0, st, other);
        CHECK(hipGraphLaunch(ex, st));
        wg += check("after-gap", rep) != 0;
    }
    printf("%s: wrong results in %d of 8 back-to-back groups and %d of 24 replays after a gap\n",
#ifdef ZERO_KERNEL
           "clear = kernel node",
#else
           "clear = memset node",
#endif
           wb, wg);
    return 0;
}
