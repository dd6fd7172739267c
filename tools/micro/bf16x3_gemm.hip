// Can three-way bf16 splits of fp32 operands (a = a1 + a2 + a3, 8 significant bits each) on the bf16 matrix cores stand in for
// the fp32 MFMA?  One wave computes C (32 x 32) = A (32 x K) . B (32 x K)^T four ways -- fp32 MFMA (v_mfma_f32_32x32x2_f32),
// split products with 6 terms (i + j <= 4), with 9 terms, and a plain bf16 product for scale -- and the host compares each
// with the fp64 product of the same fp32 inputs.  Also times the inner loops (ns per K = 16 step per wave, all SIMDs busy).
// Build: hipcc --offload-arch=gfx950 -O3 -o bf16x3_gemm bf16x3_gemm.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void split3(float a, __bf16 &h1, __bf16 &h2, __bf16 &h3)
{
    h1 = (__bf16)a;
    const float r1 = a - (float)h1;
    h2 = (__bf16)r1;
    const float r2 = r1 - (float)h2;
    h3 = (__bf16)r2;
}

template <int MODE>  // 0 fp32 mfma, 1 six terms, 2 nine terms, 3 plain bf16
__global__ void __launch_bounds__(64) gemm(int K, const float *A, const float *B, float *C)
{
    const int lane = threadIdx.x, l31 = lane & 31, h = lane >> 5;
    f32x16 acc;
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    if (MODE == 0) {
        for (int k = 0; k < K; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[l31 * K + k + h], B[l31 * K + k + h], acc, 0, 0, 0);
    } else {
        for (int k = 0; k < K; k += 16) {
            bf16x8 a1, a2, a3, b1, b2, b3;
            for (int t = 0; t < 8; ++t) {
                __bf16 x1, x2, x3;
                split3(A[l31 * K + k + h * 8 + t], x1, x2, x3);
                a1[t] = x1, a2[t] = x2, a3[t] = x3;
                split3(B[l31 * K + k + h * 8 + t], x1, x2, x3);
                b1[t] = x1, b2[t] = x2, b3[t] = x3;
            }
            // smallest terms first
            if (MODE == 2) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, b3, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b3, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, b2, acc, 0, 0, 0);
            }
            if (MODE == 1 || MODE == 2) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b3, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, b1, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b2, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b2, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b1, acc, 0, 0, 0);
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc, 0, 0, 0);
        }
    }
    for (int e = 0; e < 16; ++e) C[((e / 4) * 8 + h * 4 + (e % 4)) * 32 + l31] = acc[e];
}

// throughput: 4 independent accumulators per wave, operands in registers, 8 waves per CU
template <int MODE>
__global__ void __launch_bounds__(512) rate(float *out, int reps)
{
    const int lane = threadIdx.x & 63;
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    bf16x8 a, b;
    for (int t = 0; t < 8; ++t) a[t] = (__bf16)(0.01f * lane + t), b[t] = (__bf16)(1.f + 0.1f * t);
    float fa = 0.01f * lane, fb = 1.5f;
    for (int r = 0; r < reps; ++r) {
        if (MODE == 0) {
#pragma unroll
            for (int s = 0; s < 8; ++s) {  // one K = 16 step of four tiles
                c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, c3, 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int s = 0; s < 6; ++s) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0);
            }
        }
    }
    float s = 0.f;
    for (int e = 0; e < 16; ++e) s += c0[e] + c1[e] + c2[e] + c3[e];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

int main()
{
    const char *names[4] = {"fp32 mfma", "bf16 x3, 6 terms", "bf16 x3, 9 terms", "plain bf16"};
    for (int K : {64, 128, 256, 1024}) {
        for (int kind = 0; kind < 2; ++kind) {  // 0: post-ReLU activations x signed weights; 1: signed x signed (gradients)
            std::vector<float> A(32 * K), B(32 * K);
            srand(K + kind);
            auto rnd = [] { return (float)rand() / RAND_MAX * 2.f - 1.f; };
            for (auto &v : A) v = kind == 0 ? fmaxf(0.f, rnd() * 3.f + 0.5f) : rnd() * 1e-3f;
            for (auto &v : B) v = rnd() * 0.2f;
            std::vector<double> ref(1024, 0.0), mag(1024, 0.0);
            for (int i = 0; i < 32; ++i)
                for (int j = 0; j < 32; ++j)
                    for (int k = 0; k < K; ++k) {
                        ref[i * 32 + j] += (double)A[i * K + k] * (double)B[j * K + k];
                        mag[i * 32 + j] += fabs((double)A[i * K + k] * (double)B[j * K + k]);
                    }
            float *dA, *dB, *dC;
            hipMalloc(&dA, A.size() * 4), hipMalloc(&dB, B.size() * 4), hipMalloc(&dC, 4096);
            hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice), hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
            for (int m = 0; m < 4; ++m) {
                if (m == 0) hipLaunchKernelGGL(gemm<0>, dim3(1), dim3(64), 0, 0, K, dA, dB, dC);
                if (m == 1) hipLaunchKernelGGL(gemm<1>, dim3(1), dim3(64), 0, 0, K, dA, dB, dC);
                if (m == 2) hipLaunchKernelGGL(gemm<2>, dim3(1), dim3(64), 0, 0, K, dA, dB, dC);
                if (m == 3) hipLaunchKernelGGL(gemm<3>, dim3(1), dim3(64), 0, 0, K, dA, dB, dC);
                std::vector<float> C(1024);
                hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost);
                double emax = 0, esum = 0;  // error relative to sum |a b| (the natural scale of the rounding errors)
                for (int i = 0; i < 1024; ++i) {
                    const double e = fabs((double)C[i] - ref[i]) / mag[i];
                    emax = fmax(emax, e), esum += e;
                }
                printf("K=%4d %s  %-18s  max err / sum|ab| = %.3e   mean = %.3e\n", K, kind ? "grad-like" : "act x W  ", names[m], emax, esum / 1024);
            }
            hipFree(dA), hipFree(dB), hipFree(dC);
        }
    }
    float *out;
    hipMalloc(&out, 1024 * 512 * 4);
    for (int m = 0; m < 2; ++m) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0), hipEventCreate(&e1);
        const int reps = 2000;
        for (int w = 0; w < 2; ++w) {
            hipEventRecord(e0);
            if (m == 0) hipLaunchKernelGGL(rate<0>, dim3(1024), dim3(512), 0, 0, out, reps);
            else hipLaunchKernelGGL(rate<1>, dim3(1024), dim3(512), 0, 0, out, reps);
            hipEventRecord(e1), hipEventSynchronize(e1);
        }
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        // work per rep and wave: four 32x32 tiles x K=16 fp32-equivalent = 4 * 32*32*16*2 flop
        const double flop = 1024.0 * 8 * reps * 4 * 32 * 32 * 16 * 2;
        printf("%-28s %.3f ms  -> %.1f fp32-equivalent TFLOP/s\n", m ? "bf16 x3 (6 MFMA per K=16)" : "fp32 MFMA (8 per K=16)", ms, flop / ms * 1e-9);
    }
    return 0;
}
