// Cost of a barrier among 8 co-resident workgroups WITH a data exchange (each writes a 4 KB slab, then reads all 32 KB),
// the pattern a merged FC-chain kernel would need between layers.  Variants:
//   0: plain stores + release/acquire at agent scope (L2 write-back + invalidate)
//   1: relaxed agent-scope atomic stores / loads for the data (write-through, L2 bypass), relaxed counter, s_waitcnt only
// Compare with a kernel boundary inside a hipGraph (tools/micro/atomics_bench.hip: ~2.2 us for tiny kernels, ~4.5 us with
// the dependent first load).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int MODE>
__global__ void __launch_bounds__(256) chain_kernel(int iters, float *buf, unsigned *counter, float *out, int *err)
{
    const int nb = gridDim.x, b = blockIdx.x, t = threadIdx.x;
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        float *cur = buf + (size_t)(it & 1) * nb * 1024;
        // produce: 1024 floats per block
        for (int i = t; i < 1024; i += 256) {
            const float v = (float)(it + b + i) + acc * 1e-9f;
            if (MODE == 0) cur[b * 1024 + i] = v;
            else __hip_atomic_store(cur + b * 1024 + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (MODE == 1) __builtin_amdgcn_s_waitcnt(0);  // all stores acknowledged
        __syncthreads();
        if (t == 0) {
            const unsigned target = (unsigned)(it + 1) * nb;
            if (MODE == 0) __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            else __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int spins = 0;
            while (true) {
                const unsigned c = MODE == 0 ? __hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)
                                             : __hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (c >= target) break;
                if (++spins > 2000000) { *err = 1; break; }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
        // consume: all slabs
        for (int i = t; i < nb * 1024; i += 256) {
            if (MODE == 0) acc += cur[i];
            else acc += __hip_atomic_load(cur + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    out[b * 256 + t] = acc;
}

int main()
{
    hipStream_t st;
    CK(hipStreamCreate(&st));
    float *buf, *out; unsigned *counter; int *err;
    CK(hipMalloc(&buf, 2 * 8 * 1024 * 4)); CK(hipMalloc(&out, 8 * 256 * 4)); CK(hipMalloc(&counter, 4)); CK(hipMalloc(&err, 4));
    float *junk; CK(hipMalloc(&junk, 64 << 20));
    for (int mode = 0; mode < 2; ++mode)
        for (int dirty = 0; dirty < 2; ++dirty)
            for (int iters : {1, 11, 101}) {
                float best = 1e9f;
                for (int rep = 0; rep < 5; ++rep) {
                    CK(hipMemsetAsync(counter, 0, 4, st)); CK(hipMemsetAsync(err, 0, 4, st));
                    if (dirty) CK(hipMemsetAsync(junk, rep, 64 << 20, st));  // leave the L2s full of dirty lines
                    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
                    hipEventRecord(e0, st);
                    if (mode == 0) hipLaunchKernelGGL(chain_kernel<0>, dim3(8), dim3(256), 0, st, iters, buf, counter, out, err);
                    else hipLaunchKernelGGL(chain_kernel<1>, dim3(8), dim3(256), 0, st, iters, buf, counter, out, err);
                    hipEventRecord(e1, st); hipEventSynchronize(e1);
                    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
                    best = ms < best ? ms : best;
                }
                int h = 0; CK(hipMemcpy(&h, err, 4, hipMemcpyDeviceToHost));
                printf("mode %d dirty %d iters %3d: %.2f us total%s\n", mode, dirty, iters, best * 1e3f, h ? "  (SPIN TIMEOUT)" : "");
            }
    return 0;
}
