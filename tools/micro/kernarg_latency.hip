// What a wave pays before useful work, three probes (thread 0 of every workgroup reads the 100 MHz wall clock):
//   (a) first instruction -> a scalar kernel argument has arrived;  (b) -> a dword loaded through a pointer argument has arrived;
//   (c) 8 KB of straight-line code (2048 dependent 4-byte VALU instructions), first pass against second pass of the same wave.
// Launched back to back on a stream and as the nodes of one hipGraph, optionally behind a kernel that streams 64 MB (nothing of
// the launch before left in the L2 -- the situation inside the training step).  Stamps go to a __device__ array (PC-relative
// address: needs no argument).  Measured on MI355X: arguments +0.20 us (0.08 back to back in a graph), dependent load +0.20 us
// (0.08 with a warm L2), code 3.44 us on BOTH passes (4.03 cycles per instruction: sequential instruction fetch is hidden
// completely) -- none of the three explains why a wave's first query in the pair scan costs 1.3 us more than its second
// (DESIGN.md 4.1); what remains are data-side first touches and the dispatch ramp.
//   hipcc --offload-arch=gfx950 -O3 -o kernarg_latency kernarg_latency.hip && ./kernarg_latency
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int kLaunches = 64, kBlocks = 256;
__device__ unsigned long long g_t[kLaunches * kBlocks * 4];

struct Pad { int v[48]; };  // the step's kernels carry 200-500 bytes of arguments

__global__ void __launch_bounds__(256) probe_kernel(int launch, const int *p, int a, Pad pad, int *sink)
{
    const unsigned long long t0 = wall_clock64();
    int x = a + pad.v[47];
    asm volatile("" : "+s"(x));  // the arguments must have arrived
    const unsigned long long t1 = wall_clock64();
    int y = __builtin_amdgcn_readfirstlane(p[blockIdx.x * 64]);  // a load through a pointer argument
    asm volatile("" : "+s"(y));
    const unsigned long long t2 = wall_clock64();
    if (threadIdx.x == 0) {
        unsigned long long *o = g_t + ((size_t)launch * kBlocks + blockIdx.x) * 4;
        o[0] = t0, o[1] = t1, o[2] = t2;
        if (x + y == 0x7fffffff) *sink = x;
    }
}

// Cold instruction fetch: a straight-line stretch of 2048 four-byte VALU instructions (8 KB of code = 128 cache lines), run
// twice by the same wave.  Pass 2 finds the code in the instruction cache; the difference is what fetching it cost.
__global__ void __launch_bounds__(64) code_kernel(int launch, int *sink)
{
    int v = threadIdx.x;
    unsigned long long t[3];
    t[0] = wall_clock64();
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
        asm volatile(".rept 2048\n v_add_u32 %0, %0, %0\n .endr" : "+v"(v));
        t[pass + 1] = wall_clock64();
    }
    if (threadIdx.x == 0) {
        unsigned long long *o = g_t + ((size_t)launch * kBlocks + blockIdx.x) * 4;
        o[0] = t[0], o[1] = t[1], o[2] = t[2];
        if (v == 0x12345) *sink = v;
    }
}

__global__ void __launch_bounds__(256) flush_kernel(float4 *buf, size_t n)
{
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        float4 v = buf[i];
        v.x += 1.f;
        buf[i] = v;
    }
}

static void report(const char *name)
{
    std::vector<unsigned long long> h((size_t)kLaunches * kBlocks * 4);
    (void)hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_t), h.size() * 8);
    std::vector<double> d1, d2, first1;
    for (int l = 8; l < kLaunches; ++l) {
        unsigned long long tmin = ~0ull;
        for (int b = 0; b < kBlocks; ++b) tmin = std::min(tmin, h[((size_t)l * kBlocks + b) * 4]);
        for (int b = 0; b < kBlocks; ++b) {
            const unsigned long long *o = &h[((size_t)l * kBlocks + b) * 4];
            d1.push_back((o[1] - o[0]) / 100.0), d2.push_back((o[2] - o[1]) / 100.0);
            if (o[0] == tmin) first1.push_back((o[1] - o[0]) / 100.0);
        }
    }
    auto med = [](std::vector<double> &v, double q) { std::sort(v.begin(), v.end()); return v[(size_t)(q * (v.size() - 1))]; };
    printf("%-28s arguments arrive +%.2f us (p10 %.2f, p90 %.2f; first workgroup of a launch %.2f), dependent load +%.2f us (p10 %.2f, p90 %.2f)\n",
           name, med(d1, .5), med(d1, .1), med(d1, .9), med(first1, .5), med(d2, .5), med(d2, .1), med(d2, .9));
}

int main()
{
    int *p, *sink;
    float4 *buf;
    const size_t nflush = (64u << 20) / 16;
    CK(hipMalloc(&p, kBlocks * 64 * 4 + 64));
    CK(hipMalloc(&sink, 4));
    CK(hipMalloc(&buf, nflush * 16));
    CK(hipMemset(p, 0, kBlocks * 64 * 4 + 64));
    CK(hipMemset(buf, 0, nflush * 16));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    Pad pad{};
    for (int flush = 0; flush < 2; ++flush) {
        for (int l = 0; l < kLaunches; ++l) {
            if (flush) hipLaunchKernelGGL(flush_kernel, dim3(2048), dim3(256), 0, st, buf, nflush);
            hipLaunchKernelGGL(probe_kernel, dim3(kBlocks), dim3(256), 0, st, l, p, l, pad, sink);
        }
        CK(hipStreamSynchronize(st));
        report(flush ? "stream, L2 flushed between:" : "stream, back to back:");
        hipGraph_t g;
        hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        for (int l = 0; l < kLaunches; ++l) {
            if (flush) hipLaunchKernelGGL(flush_kernel, dim3(2048), dim3(256), 0, st, buf, nflush);
            hipLaunchKernelGGL(probe_kernel, dim3(kBlocks), dim3(256), 0, st, l, p, l, pad, sink);
        }
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int r = 0; r < 3; ++r) CK(hipGraphLaunch(ge, st));
        CK(hipStreamSynchronize(st));
        report(flush ? "graph, L2 flushed between:" : "graph, back to back:");
    }
    for (int flush = 0; flush < 2; ++flush) {
        hipGraph_t g;
        hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        for (int l = 0; l < kLaunches; ++l) {
            if (flush) hipLaunchKernelGGL(flush_kernel, dim3(2048), dim3(256), 0, st, buf, nflush);
            hipLaunchKernelGGL(code_kernel, dim3(kBlocks), dim3(64), 0, st, l, sink);
        }
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int r = 0; r < 3; ++r) CK(hipGraphLaunch(ge, st));
        CK(hipStreamSynchronize(st));
        printf("8 KB of straight-line code, one wave per CU: 'arguments' = first pass, 'dependent load' = second pass\n");
        report(flush ? "graph, L2 flushed between:" : "graph, back to back:");
    }
    return 0;
}
