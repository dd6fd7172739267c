// How fast do fp32 MFMAs issue on one SIMD of gfx950 from (a) one wave with a single dependent accumulator chain,
// (b) one wave with four independent accumulators, (c) both kinds co-resident on the SIMD (waves w and w+4 of a
// 512-thread workgroup share a SIMD), (d) as (c) with LDS fragment reads in the loops.  Reports ns per MFMA per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>  // 0: dep chain waves 0-3 only; 1: 4-acc waves 4-7 only; 2: both; 3: both + LDS reads; 4: dep chain in all 8 waves; 5: 4-acc in all 8
__global__ void __launch_bounds__(512) k(float *out, unsigned long long *tout, int reps)
{
    __shared__ float lds[8192];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 8192; i += 512) lds[i] = (float)(i & 7) * 0.125f;
    __syncthreads();
    f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
    float x = (float)lane * 0.01f, y = (float)wave * 0.1f + 1.f;
    const bool dep = (MODE == 0 || MODE == 2 || MODE == 3) ? wave < 4 : (MODE == 4);
    const bool ind = (MODE == 1 || MODE == 2 || MODE == 3) ? wave >= 4 : (MODE == 5);
    const unsigned long long t0 = wall_clock64();
    if (dep) {
        for (int r = 0; r < reps; ++r) {
#pragma unroll
            for (int s = 0; s < 64; ++s) {
                float a = x, b = y;
                if (MODE == 3) { a = lds[(s * 64 + lane) & 8191]; b = lds[(s * 64 + 4096 + lane) & 8191]; }
                a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, a0, 0, 0, 0);
            }
        }
    }
    if (ind) {
        for (int r = 0; r < reps; ++r) {
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                float a = x, b0 = y, b1 = y + 1.f, b2 = y + 2.f, b3 = y + 3.f;
                if (MODE == 3) {
                    a = lds[(s * 64 + lane) & 8191];
                    b0 = lds[(s * 64 + 1024 + lane) & 8191], b1 = lds[(s * 64 + 2048 + lane) & 8191];
                    b2 = lds[(s * 64 + 3072 + lane) & 8191], b3 = lds[(s * 64 + 5120 + lane) & 8191];
                }
                a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b2, a2, 0, 0, 0);
                a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b3, a3, 0, 0, 0);
            }
        }
    }
    const unsigned long long t1 = wall_clock64();
    float acc = 0.f;
    for (int e = 0; e < 16; ++e) acc += a0[e] + a1[e] + a2[e] + a3[e];
    out[blockIdx.x * 512 + tid] = acc;
    if (lane == 0) tout[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int MODE>
static void run(const char *name, int nblocks)
{
    float *out; unsigned long long *tout;
    hipMalloc(&out, nblocks * 512 * 4); hipMalloc(&tout, nblocks * 8 * 8);
    const int reps = 200;
    hipLaunchKernelGGL(k<MODE>, dim3(nblocks), dim3(512), 0, 0, out, tout, reps);
    hipDeviceSynchronize();
    hipLaunchKernelGGL(k<MODE>, dim3(nblocks), dim3(512), 0, 0, out, tout, reps);
    hipDeviceSynchronize();
    unsigned long long h[8];
    hipMemcpy(h, tout, 64, hipMemcpyDeviceToHost);
    printf("%-44s blocks %3d: ns per MFMA per wave:", name, nblocks);
    for (int w = 0; w < 8; ++w) printf(" %6.1f", h[w] * 10.0 / (reps * 64.0));
    printf("\n");
    hipFree(out); hipFree(tout);
}

int main()
{
    for (int nb : {1, 256}) {
        run<0>("dependent chain, waves 0-3 (1 wave/SIMD)", nb);
        run<1>("4 accumulators, waves 4-7 (1 wave/SIMD)", nb);
        run<2>("both kinds (2 waves/SIMD)", nb);
        run<3>("both kinds + LDS fragment reads", nb);
        run<4>("dependent chain in all 8 waves", nb);
        run<5>("4 accumulators in all 8 waves", nb);
    }
    return 0;
}
