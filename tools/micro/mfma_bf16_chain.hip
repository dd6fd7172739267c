// How fast does v_mfma_f32_32x32x16_bf16 issue on one SIMD of gfx950 as a function of the number of INDEPENDENT accumulator chains per
// wave (1, 2, 4) and of the waves per SIMD (1: 256 threads, 2: 512 threads -- waves w and w + 4 share a SIMD)?  Reports ns and cycles (at
// 2.4 GHz) per MFMA per SIMD.  Sized the round-6 decisions on the single-chain kernels (fused conv backward's data gradient, the wide
// pool layer): one chain per wave and two waves per SIMD does NOT reach the pipe's rate.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_bf16_chain mfma_bf16_chain.hip && ./mfma_bf16_chain
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NACC, int THREADS>
__global__ void __launch_bounds__(THREADS) k(float *out, unsigned long long *tout, int reps)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    f32x16 acc[NACC];
    for (int n = 0; n < NACC; ++n)
        for (int e = 0; e < 16; ++e) acc[n][e] = 0.f;
    bf16x8 a, b;
    for (int t = 0; t < 8; ++t) a[t] = (__bf16)((float)(lane + t) * 0.01f), b[t] = (__bf16)((float)(wave + t) * 0.1f);
    __syncthreads();
    const unsigned long long t0 = wall_clock64();
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int s = 0; s < 48 / NACC; ++s)
#pragma unroll
            for (int n = 0; n < NACC; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[n], 0, 0, 0);
    }
    const unsigned long long t1 = wall_clock64();
    float s = 0.f;
    for (int n = 0; n < NACC; ++n)
        for (int e = 0; e < 16; ++e) s += acc[n][e];
    out[blockIdx.x * THREADS + tid] = s;
    if (lane == 0) tout[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int NACC, int THREADS>
static void run(int nblocks)
{
    float *out; unsigned long long *tout;
    hipMalloc(&out, nblocks * THREADS * 4); hipMalloc(&tout, nblocks * 8 * 8);
    const int reps = 400;
    for (int i = 0; i < 2; ++i) {
        hipLaunchKernelGGL((k<NACC, THREADS>), dim3(nblocks), dim3(THREADS), 0, 0, out, tout, reps);
        hipDeviceSynchronize();
    }
    unsigned long long h[8];
    hipMemcpy(h, tout, 64, hipMemcpyDeviceToHost);
    const int wps = THREADS / 256;  // waves per SIMD
    unsigned long long mx = 0;
    for (int w = 0; w < THREADS / 64; ++w) mx = h[w] > mx ? h[w] : mx;  // (the slowest wave: partners on a SIMD may run one after the other)
    const double ns_wave = mx * 10.0 / (reps * 48.0), ns_simd = ns_wave / wps;
    printf("chains/wave %d  waves/SIMD %d  blocks %3d:  %6.1f ns per MFMA per wave, %6.1f ns per MFMA on the SIMD (= %5.1f cycles at 2.4 GHz)\n", NACC,
           wps, nblocks, ns_wave, ns_simd, ns_simd * 2.4);
    hipFree(out); hipFree(tout);
}

int main()
{
    for (int nb : {1, 256}) {
        run<1, 256>(nb); run<2, 256>(nb); run<4, 256>(nb);
        run<1, 512>(nb); run<2, 512>(nb); run<4, 512>(nb);
    }
    return 0;
}
