// step_tail.h -- the scalar tail of the sampler step's loss side (keys mode): temperature gradient from the per-workgroup
// sigma partials, the loss value from the scan's query-side partials, and the reset of the per-point key table.  Nothing on
// the step's critical path waits for these, so they can ride in ANY later launch: sigma_grad_kernel (geometry_ops.hip) runs
// them as a launch of their own; the closing kernel of the conv-stack backward (pointnet_mlp_backward.hip) runs them in two extra
// workgroups when it is handed a StepTail (sn_conv_stack_backward, engine path: one launch less).
#pragma once
#include "sn_common.h"

namespace sn {

struct StepLossKeysFinal {
    int B, G, M, N, nproj;
    float w, alpha, lmbda, min_sigma;
    const float *qpart;   // [B][G][2]  (sum dist_q, sum proj)
    const sn_u64 *qmax;   // [B][G]     max (dist_q, ~query) key
    const float *dpsum;   // [B]
    const float *temperature;
    float *loss;          // NULL: nothing to do
    sn_u64 *keys;         // [nkeys] inverted per-point keys, re-zeroed for the next step
    long long nkeys;
    int with_mean_proj;   // 1: the loss carries the stand-in task term mean(proj); 0: the task loss lives outside (its gradient
                          // arrives as an explicit grad_proj)
    // error words of the step's FC chain launches (forward, backward; NULL: none): a non-zero word -- a hand-off between
    // their workgroups timed out -- turns the loss value into NaN (sn_step_tail_set_error_words)
    const unsigned *chain_err[2];
};

struct StepTail {         // plain data: also the blob sn_sampler_step_loss_keys hands to sn_conv_stack_backward
    int nparts;           // 0: nothing to do
    const float *gsig;    // [nparts] partial d loss / d sigma
    const float *temperature;
    float min_sigma;
    float *grad_T;
    const float *grad_loss;
    float lmbda;
    StepLossKeysFinal kf;
};

// loss value (one wave): lane b carries clouds b, b + 64, ...; fixed xor tree
__device__ __forceinline__ void step_loss_keys_final(const StepLossKeysFinal &f, int t)
{
    float s1 = 0.f, mx = 0.f, s2 = 0.f, sp = 0.f;
    for (int b = t; b < f.B; b += 64) {
        float a1 = 0.f, ap = 0.f;
        sn_u64 mk = 0;
        for (int g = 0; g < f.G; ++g) {
            const size_t o = (size_t)b * f.G + g;
            a1 += f.qpart[o * 2], ap += f.qpart[o * 2 + 1];
            mk = f.qmax[o] > mk ? f.qmax[o] : mk;
        }
        s1 += a1, sp += ap, mx += key_dist(mk), s2 += f.dpsum[b];
    }
    const float T = *f.temperature;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        s1 += __shfl_xor(s1, o);
        mx += __shfl_xor(mx, o);
        s2 += __shfl_xor(s2, o);
        sp += __shfl_xor(sp, o);
    }
    if (t != 0) return;
    {
#pragma clang fp contract(off)  // (as sigma_grad_block: the same value from either translation unit)
        const float c12 = s1 / ((float)f.B * (float)f.M), cmax = mx / (float)f.B, c21 = s2 / ((float)f.B * (float)f.N);
        const float lsimp = c12 + cmax + f.w * c21;
        float L = f.alpha * lsimp + f.lmbda * sn_sigma(T, f.min_sigma);
        if (f.with_mean_proj) L = L + sp / ((float)f.B * (float)f.nproj);
        unsigned err = 0;
        if (f.chain_err[0]) err |= *f.chain_err[0];
        if (f.chain_err[1]) err |= *f.chain_err[1];
        f.loss[0] = err ? __builtin_nanf("") : L;
        f.loss[1] = lsimp;
    }
}

// d loss / dT from the per-workgroup partials of d loss / d sigma:  sigma = max(T^2, min_sigma)
//   dT = (sum partial) * 2T * [T^2 > min_sigma]   (torch.max splits the gradient evenly on an exact tie)
// First 256 threads of the calling workgroup (all of them must call: one barrier inside); red: 4 floats of LDS.
// Fixed-order reduction: strided per-thread sums, xor tree inside each wave, the four wave totals in order.
__device__ __forceinline__ void sigma_grad_block(int nparts, const float *__restrict__ partial, const float *__restrict__ temperature,
                                                 float min_sigma, float *__restrict__ grad_T,
                                                 const float *__restrict__ gsigma_direct, float direct_scale, float *red)
{
    const int t = threadIdx.x;
    const float T = *temperature;
    const float direct = gsigma_direct ? *gsigma_direct * direct_scale : 0.f;  // in flight during the reduction
    float acc = 0.f;
    if (t < 256)
        for (int i = t; i < nparts; i += 256) acc += partial[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (t < 256 && (t & 63) == 0) red[t >> 6] = acc;
    __syncthreads();
    if (t == 0) {
        // (this block is compiled into two translation units with different contraction defaults -- the geometric kernels are built
        //  with -ffp-contract=off, the MLP ones are not: product, then sum, wherever it runs)
#pragma clang fp contract(off)
        const float tot = (red[0] + red[1]) + (red[2] + red[3]);
        const float t2 = T * T;
        const float w = t2 > min_sigma ? 1.f : (t2 == min_sigma ? 0.5f : 0.f);
        // + d loss / d sigma of a term that depends on sigma directly (lmbda * sigma in the sampler step's loss)
        grad_T[0] = tot * w * 2.0f * T + direct * w * 2.0f * T;
    }
}

}  // namespace sn
