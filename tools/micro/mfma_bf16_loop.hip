// The inner loop of the split-bf16 GEMM kernels in isolation: per k-step 3 x ds_read_b128 (the next step's B fragments, double-buffered
// registers) + 6 DEPENDENT v_mfma_f32_32x32x16_bf16 (+ optional VALU fillers), 8 k-steps per "block", optionally an LDS-only barrier per
// block; 512 threads = 2 waves per SIMD.  Which ingredient takes the matrix pipe from 100 % (mfma_bf16_chain.hip) to the 62 % the wide pool
// layer reaches?   hipcc --offload-arch=gfx950 -O3 -o mfma_bf16_loop mfma_bf16_loop.hip && ./mfma_bf16_loop
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE>  // bit 0: LDS fragment reads; bit 1: VALU fillers between the groups; bit 2: barrier per block; bit 3: sched_barrier fences
__global__ void __launch_bounds__(512) k(float *out, unsigned long long *tout, int nblk)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __bf16 *Bs = reinterpret_cast<__bf16 *>(lds);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
    constexpr int PITCH = 136, PL = 64 * PITCH;
    for (int i = tid; i < 3 * PL; i += 512) Bs[i] = (__bf16)((float)(i & 15) * 0.0625f);
    __syncthreads();
    f32x16 acc, accp;
    for (int e = 0; e < 16; ++e) acc[e] = 0.f, accp[e] = 0.f;
    bf16x8 a[3][8];
    for (int p = 0; p < 3; ++p)
        for (int kk = 0; kk < 8; ++kk)
            for (int t = 0; t < 8; ++t) a[p][kk][t] = (__bf16)((float)(lane + t + kk + p) * 0.01f);
    const __bf16 *cur = Bs + ((wave >> 2) * 32 + l31) * PITCH + 8 * h;
    float m = -1e30f;
    const unsigned long long t0 = wall_clock64();
    for (int blk = 0; blk < nblk; ++blk) {
        bf16x8 b[2][3];
        for (int p = 0; p < 3; ++p) b[0][p] = (MODE & 1) ? *reinterpret_cast<const bf16x8 *>(cur + p * PL) : a[p][0];
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            if (kk + 1 < 8)
                for (int p = 0; p < 3; ++p) b[(kk + 1) & 1][p] = (MODE & 1) ? *reinterpret_cast<const bf16x8 *>(cur + p * PL + (kk + 1) * 16) : a[p][kk + 1];
            if (MODE & 8) __builtin_amdgcn_sched_barrier(0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][kk], b[kk & 1][2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2][kk], b[kk & 1][0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][kk], b[kk & 1][1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][kk], b[kk & 1][1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][kk], b[kk & 1][0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][kk], b[kk & 1][0], acc, 0, 0, 0);
            if (MODE & 8) __builtin_amdgcn_sched_barrier(0);
            if (MODE & 2) {
                m = fmaxf(m, accp[2 * kk] + 0.5f);
                m = fmaxf(m, accp[2 * kk + 1] + 0.5f);
            }
        }
        accp = acc;
        if (MODE & 4) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    const unsigned long long t1 = wall_clock64();
    float s = m;
    for (int e = 0; e < 16; ++e) s += acc[e];
    out[blockIdx.x * 512 + tid] = s;
    if (lane == 0) tout[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int MODE>
static void run(const char *name, int nblocks)
{
    float *out; unsigned long long *tout;
    hipMalloc(&out, nblocks * 512 * 4); hipMalloc(&tout, nblocks * 8 * 8);
    const int nblk = 400;
    hipFuncSetAttribute((const void *)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 64 * 136 * 2);
    for (int i = 0; i < 2; ++i) {
        hipLaunchKernelGGL((k<MODE>), dim3(nblocks), dim3(512), 3 * 64 * 136 * 2, 0, out, tout, nblk);
        hipDeviceSynchronize();
    }
    unsigned long long h[8], mx = 0;
    hipMemcpy(h, tout, 64, hipMemcpyDeviceToHost);
    for (int w = 0; w < 8; ++w) mx = h[w] > mx ? h[w] : mx;
    const double ns = mx * 10.0 / (nblk * 48.0) / 2.0;
    printf("%-58s blocks %3d: %5.1f ns per MFMA on the SIMD (pipe rate under load ~16.5)\n", name, nblocks, ns);
    hipFree(out); hipFree(tout);
}

int main()
{
    for (int nb : {1, 256}) {
        run<0>("MFMAs only (operands in registers)", nb);
        run<1>("+ LDS fragment reads", nb);
        run<9>("+ LDS fragment reads, sched_barrier fences", nb);
        run<3>("+ LDS reads + VALU fillers", nb);
        run<11>("+ LDS reads + VALU fillers, fences", nb);
        run<5>("+ LDS reads + barrier per 48", nb);
        run<15>("+ LDS reads + fillers + barrier, fences (the wide kernel)", nb);
    }
    return 0;
}
