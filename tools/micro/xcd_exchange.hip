// Cost of one "layer seam" of a merged FC-head kernel: 8 workgroups each publish a 4 KB slab (32 rows x 32 columns of the
// layer's output), then every workgroup reads all 32 KB.  Unlike grid_barrier.hip (4-byte atomics, workgroups spread over the
// 8 XCDs) this follows MI355X_MICROARCH.md's price-list advice: 16-byte write-through (sc1) stores, drained, one relaxed
// arrival counter, sc1 payload loads with 8 x 16 B in flight per lane -- with the 8 workgroups either pinned to ONE XCD
// (grid of 64, blocks b % 8 == 0 work) or spread (grid of 8).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void store_sc1(f4 *p, f4 v) { asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 4" ::"v"(p), "v"(v) : "memory"); }

template <bool PINNED>
__global__ void __launch_bounds__(256) seam_kernel(int iters, f4 *buf, unsigned *counter, float *out, int *err)
{
    int b = blockIdx.x;
    if (PINNED) {
        if (b & 7) return;
        b >>= 3;
    }
    const int t = threadIdx.x;
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        f4 *cur = buf + (size_t)(it & 1) * 8 * 256;
        f4 v = {(float)(it + b + t) + acc * 1e-9f, 1.f, 2.f, 3.f};
        store_sc1(cur + b * 256 + t, v);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t == 0) {
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = (unsigned)(it + 1) * 8;
            int spins = 0;
            while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                if (++spins > 4000000) { *err = 1; break; }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
        f4 r[8];
#pragma unroll
        for (int i = 0; i < 8; ++i)
            asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=&v"(r[i]) : "v"(cur + i * 256 + t) : "memory");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < 8; ++i) acc += r[i].x + r[i].y;
    }
    out[b * 256 + t] = acc;
}

// The same seam WITHOUT drain / arrival counter / poll: the slab granules start out as a sentinel bit pattern, a consumer
// simply re-loads its 8 granules until none of them holds the sentinel any more (all four dwords checked: no assumption that a
// 16-byte store is observed untorn).  Four slabs in rotation; in iteration j a workgroup re-arms its own region of slab
// (j + 2) % 4 together with the data store (the consume phase's vmcnt(0) then covers it before the workgroup's next data leaves).
template <int MODE>
__global__ void __launch_bounds__(256) seam_sentinel_kernel(int iters, f4 *buf, float *out, int *err)
{
    int b = blockIdx.x;
    if (b & 7) return;
    b >>= 3;
    const int t = threadIdx.x;
    const unsigned S = 0xFFFFFFFFu;
    const f4 sent = {__uint_as_float(S), __uint_as_float(S), __uint_as_float(S), __uint_as_float(S)};
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        f4 *cur = buf + (size_t)(it & 3) * 8 * 256, *arm = buf + (size_t)((it + 2) & 3) * 8 * 256;
        f4 v = {(float)(it + b + t) + acc * 1e-9f, 1.f, 2.f, 3.f};
        store_sc1(cur + b * 256 + t, v);
        store_sc1(arm + b * 256 + t, sent);
        f4 r[8];
        unsigned pending = 0xffu;
        int spins = 0;
        while (pending) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (pending & (1u << i)) {
                    if (MODE == 2) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=&v"(r[i]) : "v"(cur + i * 256 + t) : "memory");
                    else asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=&v"(r[i]) : "v"(cur + i * 256 + t) : "memory");
                }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (MODE == 1) asm volatile("buffer_inv sc1" ::: "memory");
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const bool bad = __float_as_uint(r[i].x) == S || __float_as_uint(r[i].y) == S || __float_as_uint(r[i].z) == S ||
                                 __float_as_uint(r[i].w) == S;
                if (!bad) pending &= ~(1u << i);
            }
            if (++spins > 20000) { *err = 1; break; }
        }
        if (r[b].x != (float)(it + b + t) + acc * 1e-9f && 0) *err = 2;
#pragma unroll
        for (int i = 0; i < 8; ++i) acc += r[i].x + r[i].y;
    }
    out[b * 256 + t] = acc;
}

int main()
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    hipStream_t st;
    CK(hipStreamCreate(&st));
    f4 *buf; float *out; unsigned *counter; int *err;
    CK(hipMalloc(&buf, 2 * 8 * 256 * 16)); CK(hipMalloc(&out, 8 * 256 * 4)); CK(hipMalloc(&counter, 4)); CK(hipMalloc(&err, 4));
    for (int pinned = 0; pinned < 2; ++pinned)
        for (int iters : {1, 11, 101}) {
            float best = 1e9f;
            for (int rep = 0; rep < 7; ++rep) {
                CK(hipMemsetAsync(counter, 0, 4, st)); CK(hipMemsetAsync(err, 0, 4, st));
                hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
                hipEventRecord(e0, st);
                if (pinned) hipLaunchKernelGGL(seam_kernel<true>, dim3(64), dim3(256), 0, st, iters, buf, counter, out, err);
                else hipLaunchKernelGGL(seam_kernel<false>, dim3(8), dim3(256), 0, st, iters, buf, counter, out, err);
                hipEventRecord(e1, st); hipEventSynchronize(e1);
                float ms = 0; hipEventElapsedTime(&ms, e0, e1);
                best = ms < best ? ms : best;
            }
            int h = 0; CK(hipMemcpy(&h, err, 4, hipMemcpyDeviceToHost));
            printf("pinned %d iters %3d: %.2f us total%s\n", pinned, iters, best * 1e3f, h ? "  (SPIN TIMEOUT)" : "");
        }
    // sentinel hand-off (pinned): slabs armed once by the host
    f4 *buf4; CK(hipMalloc(&buf4, 4 * 8 * 256 * 16));
    float ref101 = 0.f;
    setvbuf(stdout, nullptr, _IONBF, 0);
    for (int mode = 0; mode < 3; ++mode)
    for (int iters : {1, 11, 101}) {
        float best = 1e9f;
        for (int rep = 0; rep < 7; ++rep) {
            CK(hipMemsetAsync(buf4, 0xFF, 4 * 8 * 256 * 16, st)); CK(hipMemsetAsync(err, 0, 4, st));
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0, st);
            if (mode == 0) hipLaunchKernelGGL(seam_sentinel_kernel<0>, dim3(64), dim3(256), 0, st, iters, buf4, out, err);
            else if (mode == 1) hipLaunchKernelGGL(seam_sentinel_kernel<1>, dim3(64), dim3(256), 0, st, iters, buf4, out, err);
            else hipLaunchKernelGGL(seam_sentinel_kernel<2>, dim3(64), dim3(256), 0, st, iters, buf4, out, err);
            int hh = 0; CK(hipMemcpy(&hh, err, 4, hipMemcpyDeviceToHost));
            if (hh) { rep = 7; }
            hipEventRecord(e1, st); hipEventSynchronize(e1);
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            best = ms < best ? ms : best;
        }
        int h = 0; CK(hipMemcpy(&h, err, 4, hipMemcpyDeviceToHost));
        float o = 0; CK(hipMemcpy(&o, out, 4, hipMemcpyDeviceToHost));
        if (iters == 101) ref101 = o;
        printf("sentinel mode %d pinned iters %3d: %.2f us total%s  (out[0] = %.1f)\n", mode, iters, best * 1e3f, h ? "  (SPIN TIMEOUT / MISMATCH)" : "", o);
    }
    // cross-check of the sums against the counter version on the same schedule
    CK(hipMemsetAsync(counter, 0, 4, st));
    hipLaunchKernelGGL(seam_kernel<true>, dim3(64), dim3(256), 0, st, 101, buf, counter, out, err);
    float o2 = 0; CK(hipMemcpy(&o2, out, 4, hipMemcpyDeviceToHost));
    printf("counter version out[0] = %.1f  sentinel version %.1f  %s\n", o2, ref101, o2 == ref101 ? "EQUAL" : "DIFFERENT");
    return 0;
}
