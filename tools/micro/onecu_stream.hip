// How fast can ONE workgroup (one CU) stream a weight matrix from L2 / Infinity Cache / HBM?  Sizes the single-CU FC-head idea.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(1024) stream1(const float4 *__restrict__ p, int n4, float *out)
{
    float4 acc = make_float4(0, 0, 0, 0);
    for (int i = threadIdx.x; i < n4; i += 1024 * 4) {
        float4 a = p[i], b = (i + 1024 < n4) ? p[i + 1024] : acc, c = (i + 2048 < n4) ? p[i + 2048] : acc, d = (i + 3072 < n4) ? p[i + 3072] : acc;
        acc.x += a.x + b.x + c.x + d.x; acc.y += a.y + b.y + c.y + d.y; acc.z += a.z + b.z; acc.w += c.w + d.w;
    }
    out[threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}
__global__ void thrash(float *p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = p[i] * 1.0001f + 1.f; }
int main()
{
    const int bytes = 852 * 1024, n4 = bytes / 16;
    float4 *w; float *out, *junk;
    hipMalloc(&w, bytes); hipMemset(w, 0, bytes); hipMalloc(&out, 4096);
    const int nj = 128 << 20; hipMalloc(&junk, (size_t)nj * 4); hipMemset(junk, 0, (size_t)nj * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 2; ++mode) {
        float tot = 0;
        for (int r = 0; r < 10; ++r) {
            if (mode == 1) hipLaunchKernelGGL(thrash, dim3(nj / 256), dim3(256), 0, 0, junk, nj);  // evict L2 / MALL (512 MB pass)
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(stream1, dim3(1), dim3(1024), 0, 0, w, n4, out);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (r >= 2) tot += ms;
        }
        printf("%s: one workgroup streams 852 KB in %.2f us (%.1f GB/s) incl. launch\n", mode ? "cold (after a 512 MB pass)" : "warm (L2)", tot / 8 * 1e3, bytes / (tot / 8 * 1e-3) / 1e9);
    }
    return 0;
}
