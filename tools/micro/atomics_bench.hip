// Micro-benchmarks that size design decisions for the BatchNorm-statistics path (debug tool, not part of the library):
//   * cost of a dependent kernel boundary (stream vs hipGraph)
//   * throughput of contended 64-bit integer / fp64 / fp32 atomics from 512 workgroups onto C channels x NC copies
//   * tail cost of the "last workgroup finalises" pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void empty_kernel(int *p) { if (p && threadIdx.x == 9999) p[0] = 1; }
__global__ void touch_kernel(float *p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] += 1.f; }

template <int MODE>  // 0: u64 x4, 1: f64 x2, 2: f32 x2, 3: plain partial stores (baseline)
__global__ void __launch_bounds__(256) stat_kernel(int C, int NC, unsigned long long *acc, double *accd, float *accf, float *part)
{
    const int t = threadIdx.x;
    if (t >= C) return;
    const int copy = blockIdx.x % NC;
    const float v = (float)(blockIdx.x * 131 + t) * 1e-3f;
    if (MODE == 0) {
        unsigned long long *a = acc + ((size_t)copy * C + t) * 4;
        atomicAdd(a + 0, (unsigned long long)(v * 256.f));
        atomicAdd(a + 1, (unsigned long long)(v * 65536.f));
        atomicAdd(a + 2, (unsigned long long)(v * v * 256.f));
        atomicAdd(a + 3, (unsigned long long)(v * v * 65536.f));
    } else if (MODE == 1) {
        double *a = accd + ((size_t)copy * C + t) * 2;
        atomicAdd(a + 0, (double)v);
        atomicAdd(a + 1, (double)v * v);
    } else if (MODE == 2) {
        float *a = accf + ((size_t)copy * C + t) * 2;
        atomicAdd(a + 0, v);
        atomicAdd(a + 1, v * v);
    } else {
        part[((size_t)blockIdx.x * 2 + 0) * C + t] = v;
        part[((size_t)blockIdx.x * 2 + 1) * C + t] = v * v;
    }
}

// u64 atomics + last block finalises (reads NC copies, writes coef, re-zeroes)
__global__ void __launch_bounds__(256) stat_last_kernel(int C, int NC, unsigned long long *acc, unsigned *counter, float *coef)
{
    const int t = threadIdx.x;
    __shared__ int last;
    const int copy = blockIdx.x % NC;
    const float v = (float)(blockIdx.x * 131 + t) * 1e-3f;
    if (t < C) {
        unsigned long long *a = acc + ((size_t)copy * C + t) * 4;
        atomicAdd(a + 0, (unsigned long long)(v * 256.f));
        atomicAdd(a + 1, (unsigned long long)(v * 65536.f));
        atomicAdd(a + 2, (unsigned long long)(v * v * 256.f));
        atomicAdd(a + 3, (unsigned long long)(v * v * 65536.f));
    }
    __threadfence();
    __syncthreads();
    if (t == 0) last = (atomicAdd(counter, 1u) == gridDim.x - 1);
    __syncthreads();
    if (!last) return;
    __threadfence();
    if (t < C) {
        unsigned long long s[4] = {0, 0, 0, 0};
        for (int c = 0; c < NC; ++c)
            for (int k = 0; k < 4; ++k) {
                unsigned long long *a = acc + ((size_t)c * C + t) * 4 + k;
                s[k] += __atomic_load_n(a, __ATOMIC_RELAXED);
                *a = 0;
            }
        coef[t] = (float)((double)s[0] / 256.0 + (double)s[1] / 65536.0);
        coef[C + t] = (float)((double)s[2] / 256.0 + (double)s[3] / 65536.0);
    }
    if (t == 0) *counter = 0;
}

template <class F>
static float time_us(F f, int reps, hipStream_t st)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i) f();
    hipEventRecord(e0, st);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(e1, st);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f / reps;
}

int main()
{
    hipStream_t st;
    CK(hipStreamCreate(&st));
    const int C = 128, NB = 512;
    unsigned long long *acc; double *accd; float *accf, *part, *coef; unsigned *counter; float *buf;
    CK(hipMalloc(&acc, 64 * C * 4 * 8)); CK(hipMemset(acc, 0, 64 * C * 4 * 8));
    CK(hipMalloc(&accd, 64 * C * 2 * 8)); CK(hipMemset(accd, 0, 64 * C * 2 * 8));
    CK(hipMalloc(&accf, 64 * C * 2 * 4)); CK(hipMemset(accf, 0, 64 * C * 2 * 4));
    CK(hipMalloc(&part, (size_t)NB * 4 * 2 * C * 4));
    CK(hipMalloc(&coef, 2 * C * 4));
    CK(hipMalloc(&counter, 4)); CK(hipMemset(counter, 0, 4));
    CK(hipMalloc(&buf, 1 << 24));
    printf("empty kernel, stream back-to-back: %.2f us\n", time_us([&] { hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, st, (int *)nullptr); }, 2000, st));
    printf("touch 4MB kernel, stream back-to-back: %.2f us\n", time_us([&] { hipLaunchKernelGGL(touch_kernel, dim3(4096), dim3(256), 0, st, buf, 1 << 20); }, 1000, st));
    for (int len : {50}) {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
        for (int i = 0; i < len; ++i) hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, st, (int *)nullptr);
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        printf("graph of %d empty kernels: %.2f us per kernel\n", len, time_us([&] { hipGraphLaunch(ge, st); }, 200, st) / len);
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
        for (int i = 0; i < len; ++i) hipLaunchKernelGGL(touch_kernel, dim3(4096), dim3(256), 0, st, buf, 1 << 20);
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        printf("graph of %d touch-4MB kernels: %.2f us per kernel\n", len, time_us([&] { hipGraphLaunch(ge, st); }, 200, st) / len);
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
        for (int i = 0; i < len; ++i) hipLaunchKernelGGL(touch_kernel, dim3(16), dim3(256), 0, st, buf, 4096);
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        printf("graph of %d touch-16KB kernels: %.2f us per kernel\n", len, time_us([&] { hipGraphLaunch(ge, st); }, 200, st) / len);
    }
    for (int nb : {512, 1024})
        for (int NC : {1, 4, 16, 64}) {
            float a = time_us([&] { hipLaunchKernelGGL(stat_kernel<0>, dim3(nb), dim3(256), 0, st, C, NC, acc, accd, accf, part); }, 200, st);
            float b = time_us([&] { hipLaunchKernelGGL(stat_kernel<1>, dim3(nb), dim3(256), 0, st, C, NC, acc, accd, accf, part); }, 200, st);
            float c = time_us([&] { hipLaunchKernelGGL(stat_kernel<2>, dim3(nb), dim3(256), 0, st, C, NC, acc, accd, accf, part); }, 200, st);
            float d = time_us([&] { hipLaunchKernelGGL(stat_kernel<3>, dim3(nb), dim3(256), 0, st, C, NC, acc, accd, accf, part); }, 200, st);
            CK(hipMemsetAsync(acc, 0, 64 * C * 4 * 8, st));
            float e = time_us([&] { hipLaunchKernelGGL(stat_last_kernel, dim3(nb), dim3(256), 0, st, C, NC, acc, counter, coef); }, 200, st);
            printf("blocks %4d copies %2d: u64x4 %.2f us  f64x2 %.2f us  f32x2 %.2f us  stores %.2f us  u64x4+last-block %.2f us\n", nb, NC, a, b, c, d, e);
        }
    return 0;
}
