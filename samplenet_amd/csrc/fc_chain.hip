// fc_chain.hip -- the FC head of the sampler (registration/src/samplenet.py:53-59, :97-104: 3 x [Linear -> BatchNorm1d -> ReLU],
// rows R <= 32) as ONE resident-workgroup launch per direction: the workgroups of a layer keep running and hand activations /
// gradients to each other through write-through stores and arrival counters instead of ending the kernel after every layer.
// Arithmetic is that of the layer-by-layer R <= 32 kernels in pointnet_mlp.hip (bit-identical results).
#include "mlp_device.h"

namespace sn {

// ------------------------------------------------------------------------------------------------
// FC head forward as ONE launch (rows R <= 32, hidden width H, nl BatchNorm + ReLU layers): the H / 32 workgroups of a layer
// keep running and hand the layer's activations to each other through HBM instead of ending the kernel after every layer --
// a dependent launch of an 8-workgroup kernel costs 6-9 us here (launch boundary + cold operand fetch + store drain), a seam
// 1.2-1.8 us (tools/micro/xcd_exchange.hip).  Per workgroup: every layer's 32-column weight slice is fetched into LDS at
// the START (all fetches in flight at once, none behind a dependency); per layer: MFMA over K split across the 4 waves ->
// wave 0: bias, two-pass BatchNorm statistics (all rows are local), coefficients, pre-BN tile -> every thread publishes
// 16 bytes of the ACTIVATED 32 x 32 tile with a write-through (sc1) store -> drained -> one arrival atomic -> poll -> every
// thread gathers the full 32 x H activation with 8 sc1 16-byte loads in flight (MI355X_MICROARCH.md, inter-workgroup
// visibility: sc1 stores + sc1 loads need no fence).  Workgroups sit on ONE XCD (grid of 8 x H/32, blocks with b % 8 != 0
// exit: observed placement b % 8 -> XCD, a speed matter only).  Arrival counters are monotonic over launches: the epoch word
// is read by every workgroup before its first arrival and advanced by workgroup 0 after the first seam (no reset, no host
// involvement: safe under graph replay).  A poll that exceeds its bound sets the error word instead of hanging the GPU.
// Arithmetic (K split, summation order, BatchNorm expressions) is that of small_fwd_lds_kernel: results are bit-identical
// to the layer-by-layer launches.
// ------------------------------------------------------------------------------------------------
constexpr int kFcChainMaxLayers = 4;
struct FcChainLayer {
    const float *W, *bias, *gamma, *beta;
    float *running_mean, *running_var;
    long long *num_batches_tracked;
    float *z, *coef;  // outputs: pre-BN (R, H) and (4, H)
    float eps, momentum;
};
// POOL variant: the last conv layer's BatchNorm finalisation + max-pool pick (bn_finalize_pool_kernel) as stage -1 of the chain
struct FcChainPool {
    long long *acc;        // fixed-point statistics of the last conv layer (cleared here: this launch is their only reader)
    long long *zero_ptr;   // the accumulators the PREVIOUS kernel consumed
    int zero_n;
    const unsigned long long *keys;  // [R][2][C0] (max Z, first row) / (min Z, first row) keys left by the last conv layer
    BnFwd bn;
    float *pooled, *zsel;  // (R, C0)
    int *argsel;
};
// OUT variant: the head's OUTPUT layer behind the hidden layers -- Linear (Co x H) + BatchNorm WITHOUT activation, the classification
// task's sampler (classification/models/samplenet_model.py:100-108) -- as one more stage of the chain: the last hidden layer hands its
// activations over like the others, workgroups 0 .. Co / 32 - 1 then multiply their 32-column slice (weights fetched up front with
// everything else, parked in registers, staged into the LDS slot of a hidden layer that is done) and finish the BatchNorm of their
// columns (all R rows are local): z (R, Co) pre-BN, coef (4, Co), y (R, Co) = z scale + shift, running statistics.
struct FcChainOut {
    const float *W, *bias, *gamma, *beta;
    float *running_mean, *running_var;
    long long *num_batches_tracked;
    float *z, *coef, *y;
    float eps, momentum;
    int Co;
};
struct FcChainArgs {
    const float *a0;  // (R, C0): input of the first layer, used as is (pooled features)
    int R, C0, H, nl;
    FcChainPool P;
    FcChainOut O;
    FcChainLayer L[kFcChainMaxLayers];
    float *xbuf;     // [2][32][H] exchange slabs
    unsigned *sync;  // [0] epoch, [1 + s] arrivals at seam s, [15] error flag -- persistent, zero-initialised once
    double rinv_rows, unbias;  // 1 / R and R / (R - 1) (1 when R == 1)
};

// sum over the 8 lanes of a column group (lanes 8 c .. 8 c + 7), every lane receives the total: xor 1, xor 2 inside the quad,
// then the mirrored lane of the other quad (which holds that quad's total)
__device__ __forceinline__ float sum8_dpp(float x)
{
    x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0xB1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
    x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x4E, 0xf, 0xf, true));   // quad_perm [2,3,0,1]
    x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x141, 0xf, 0xf, true));  // row_half_mirror
    return x;
}

// Sum of a column's 32 rows in the ORDER of the one-wave epilogue this replaces (small_fwd_lds_kernel, bit for bit): there a
// lane held the 16 rows of its half (rows 8 g + 4 h + 0..3, g = 0..3) and added them in register order, then the two halves
// were added.  Here the rows 4 rg .. 4 rg + 3 of lane rg belong to half h = rg & 1, group g = rg >> 1: the running sum of a half
// walks over its four lanes (rg = h, h + 2, h + 4, h + 6) by DPP row_shr:2, each adding its four rows in order; the two ends
// (rg = 6, 7) are added and handed to all 8 lanes.  v[i] must already be 0 for rows that do not exist.
__device__ __forceinline__ float col_sum_seq(const float (&v)[4], int rg)
{
    float a = 0.f;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float in = g == 0 ? 0.f : __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a), 0x112, 0xf, 0xf, true));  // row_shr:2
        if ((rg >> 1) == g) a = (((in + v[0]) + v[1]) + v[2]) + v[3];
    }
    return sum8_dpp(rg >= 6 ? a : 0.f);  // = end(h = 0) + end(h = 1); the other lanes contribute exact zeros
}

typedef float f32x4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_sc1_b128(float *p, f32x4v v)
{
    // s_nop: a VMEM store of more than 8 bytes still reads its upper data registers for a few cycles after issue; the compiler
    // pads that hazard for its own instructions but cannot see into inline asm (observed: bytes 8..15 of the store corrupted in
    // the lanes whose data registers the next VALU instruction rewrote)
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 4" ::"v"(p), "v"(v) : "memory");
}

// C0T / NLT > 0: input width / layer count known at compile time (the sampler's 128 -> 256 x 3 head): the operand fetches then
// unroll into ONE batch of loads; 0: run-time loops.
// One seam of the chain kernels: arrive at counter `ctr`, wait until all nwg workgroups of this launch have (counters are
// monotonic over launches: target = (epoch + 1) * nwg).  Called by thread 0 between two lds_barrier().
// Returns true when the poll gave up (never on a healthy run: the workgroups of a chain launch must be resident together --
// 8 or 16 workgroups of ~137 KB LDS on otherwise free CUs; a device kept full by other work can starve one of them).  The
// caller then poisons its outputs with NaN, and the error word stays set for the loss tail / the host (sn_fc_chain_error).
// Poll bound: sync[13] when non-zero (tests), else 2^22 polls (seconds).
constexpr int kFcChainPolls = 1 << 22;
// the words of a chain launch's `sync` state sit 128 bytes apart (word i at sync[i * kFcSyncStride]): epoch, the per-seam
// arrival counters, the poll bound and the error word each own a cache line -- 8..16 workgroups add to and poll different
// counters at the same time, and on ONE line every poll queues behind the others' atomics
constexpr int kFcSyncStride = SN_FC_SYNC_STRIDE;
__device__ __forceinline__ bool fc_chain_seam(unsigned *sync, int ctr, unsigned epoch, int nwg, unsigned errcode, int limit)
{
    __hip_atomic_fetch_add(sync + ctr * kFcSyncStride, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned target = (epoch + 1u) * (unsigned)nwg;
    int spins = 0;
    while ((int)(__hip_atomic_load(sync + ctr * kFcSyncStride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
        if (++spins > limit) {  // report instead of hanging the device
            __hip_atomic_store(sync + 15 * kFcSyncStride, errcode, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return true;
        }
        __builtin_amdgcn_s_sleep(1);
    }
    return false;
}

template <int C0T, int NLT, bool POOL = false, bool OUT = false>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) fc_chain_fwd_kernel(FcChainArgs g)
{
    static_assert(!OUT || (POOL && NLT >= 2), "output stage: the pooled 128 -> 256 x 3 head");
    extern __shared__ __attribute__((aligned(16))) float sm[];
    __shared__ unsigned s_epoch, s_limit, s_bad;
    if (blockIdx.x & 7) return;
    const int wg = blockIdx.x >> 3, nwg = g.H / 32;
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int R = g.R, H = g.H, C0 = C0T > 0 ? C0T : g.C0, nl = NLT > 0 ? NLT : g.nl;
    const int LDA = (C0 > H ? C0 : H) + 4;
    float *As = sm;                 // [32][LDA]
    float *W0s = As + 32 * LDA;     // [32][C0 + 4]            weight slice of layer 0
    float *Whs = W0s + 32 * (C0 + 4);  // [nl - 1][32][H + 4]  weight slices of layers 1 ..
    float *red = Whs + (size_t)(nl - 1) * 32 * (H + 4);  // [4][64][kRsPitch]: the waves' K partials (wave_reduce_scatter4)
    float *Ts = red + kRsFloats;                                           // [32][36] pre-BN tile
    float *Ta = Ts + 32 * 36;                                              // [32][36] activated tile
    const int col0 = wg * 32;
    FC_TL(0, wg, 0);
    if (tid == 0) {
        const unsigned lim = g.sync[13 * kFcSyncStride];
        s_epoch = __hip_atomic_load(g.sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_limit = lim ? lim : (unsigned)kFcChainPolls, s_bad = 0;
    }

    // ---- every layer's weight slice + the first operand: all fetches issued up front, staged into LDS as they land
    f32x4v wo[OUT ? 8 : 1];  // (OUT: the output layer's 32 x H slice, 4 rows per pass as the hidden slices)
    if constexpr (POOL) {
        // ---- stage -1: BatchNorm of the last conv layer from its fixed-point sums + the max-pool pick.  The producing layer
        // left, per cloud and channel, (max Z, first row) and (min Z, first row) as 64-bit keys (FwdArgs::pool_keys); EVERY
        // workgroup finalises all C0 channels and decodes all R x C0 pooled features itself (32 KB of keys + 35 words of sums
        // per channel, all requested up front together with the weight slices) -- no exchange, no seam in front of fc1.  The
        // workgroup that owns a channel (16 per workgroup) stores its coefficients, running statistics and the pooled / argsel /
        // zsel rows for the backward, and clears the sums after the first layer's seam (every workgroup has read them by then).
        constexpr int CP = C0T > 0 ? C0T : 128, NH = NLT > 1 ? NLT - 1 : 1;
        static_assert(CP == 128, "pool stage: 128 pooled channels");
        const FcChainPool &P = g.P;
        const int pc = tid & 127;                    // channel whose coefficients this thread computes (two threads per channel)
        const FxRaw2 fx = fx_load2(P.acc, pc);
        const BnFwdIn in{P.bn.gamma[pc], P.bn.beta[pc], P.bn.running_mean[pc], P.bn.running_var[pc]};  // (host: never NULL here)
        // keys: thread -> cloud kb = tid >> 3, channels kc0 = 16 (tid & 7) .. + 15, both selections
        const int kb = tid >> 3, kc0 = (tid & 7) * 16;
        typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
        u64x2 kmx[8], kmn[8];
        {
            const unsigned long long *kp = P.keys + ((size_t)min(kb, R - 1) * 2) * CP + kc0;
#pragma unroll
            for (int q = 0; q < 8; ++q) kmx[q] = *reinterpret_cast<const u64x2 *>(kp + 2 * q), kmn[q] = *reinterpret_cast<const u64x2 *>(kp + CP + 2 * q);
        }
        constexpr int q4 = CP / 4, rpp = 256 / q4, npass = 32 / rpp;
        const int c4 = (tid % q4) * 4, r0 = tid / q4;
        const int hc4 = (tid % 64) * 4, hr0 = tid / 64;
        f32x4v wv[npass], wh[NH][8];  // (native vectors: arrays of the float4 STRUCT end up in scratch across the asm)
#pragma unroll
        for (int q = 0; q < npass; ++q) wv[q] = *reinterpret_cast<const f32x4v *>(g.L[0].W + (size_t)(col0 + r0 + q * rpp) * C0 + c4);
#pragma unroll
        for (int l = 1; l < NLT; ++l)
#pragma unroll
            for (int q = 0; q < 8; ++q) wh[l - 1][q] = *reinterpret_cast<const f32x4v *>(g.L[l].W + (size_t)(col0 + hr0 + q * 4) * H + hc4);
        if constexpr (OUT) {  // the output layer's slice (rows clamped: workgroups past its width fetch a valid row they never use)
#pragma unroll
            for (int q = 0; q < 8; ++q) wo[q] = *reinterpret_cast<const f32x4v *>(g.O.W + (size_t)min(col0 + hr0 + q * 4, g.O.Co - 1) * H + hc4);
        }
        // every fetch of the kernel is in flight now; nothing below may be hoisted between them
        asm volatile("" ::: "memory");
        double s, ss;
        fx_total2<kFxShiftFwd>(fx, s, ss);
        FC_TL(0, wg, 24);
        const bool owner = tid < 128 && (pc >> 4) == wg;
        const float2 cf = bn_finalize_channel(P.bn, CP, pc, s, ss, in, owner);
        if (wg == 0 && tid == 0 && P.bn.num_batches_tracked) *P.bn.num_batches_tracked += 1;
        float *cfs = Ta;  // [2][128] scale | shift (the tile scratch is idle until layer 0's epilogue)
        if (tid < 128) cfs[pc] = cf.x, cfs[CP + pc] = cf.y;
        lds_barrier();
        FC_TL(0, wg, 25);
        {
            float pooled[16], zs[16];
            int ar[16];
#pragma unroll
            for (int q = 0; q < 8; ++q)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int c = kc0 + 2 * q + t;
                    const float sc = cfs[c], sh = cfs[CP + c];
                    float v;
                    int row;
                    if (sc >= 0.f) {
                        pool_key_decode(kmx[q][t], v, row);
                    } else {
                        pool_key_decode(kmn[q][t], v, row);
                        v = -v;
                    }
                    zs[2 * q + t] = v, ar[2 * q + t] = row;
                    pooled[2 * q + t] = kb < R ? relu_np(fmaf(v, sc, sh)) : 0.f;
                }
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *reinterpret_cast<float4 *>(As + kb * LDA + kc0 + 4 * q) = make_float4(pooled[4 * q], pooled[4 * q + 1], pooled[4 * q + 2], pooled[4 * q + 3]);
            if ((tid & 7) == wg && kb < R) {  // this workgroup's 16 channels of cloud kb: the rows the backward reads
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const size_t o = (size_t)kb * CP + kc0 + 4 * q;
                    *reinterpret_cast<float4 *>(P.pooled + o) = make_float4(pooled[4 * q], pooled[4 * q + 1], pooled[4 * q + 2], pooled[4 * q + 3]);
                    *reinterpret_cast<float4 *>(P.zsel + o) = make_float4(zs[4 * q], zs[4 * q + 1], zs[4 * q + 2], zs[4 * q + 3]);
                    *reinterpret_cast<int4 *>(P.argsel + o) = make_int4(ar[4 * q], ar[4 * q + 1], ar[4 * q + 2], ar[4 * q + 3]);
                }
            }
        }
        FC_TL(0, wg, 26);
#pragma unroll
        for (int q = 0; q < npass; ++q) *reinterpret_cast<f32x4v *>(W0s + (r0 + q * rpp) * (C0 + 4) + c4) = wv[q];
#pragma unroll
        for (int l = 1; l < NLT; ++l)
#pragma unroll
            for (int q = 0; q < 8; ++q)
                *reinterpret_cast<f32x4v *>(Whs + (size_t)(l - 1) * 32 * (H + 4) + (hr0 + q * 4) * (H + 4) + hc4) = wh[l - 1][q];
        FC_TL(0, wg, 27);
        fx_clear_share(P.zero_ptr, P.zero_n, wg, nwg, tid, 256);
    } else if constexpr (C0T > 0 && NLT > 0) {
        // compile-time shape: every load of the kernel's operands is issued before the first LDS write (no loop-carried
        // load -> store dependencies, no branches around loads), layer 0's operands first
        constexpr int q4 = (C0T > 0 ? C0T : 128) / 4, rpp = 256 / q4, npass = 32 / rpp;
        constexpr int NH = NLT > 1 ? NLT - 1 : 1;
        const int c4 = (tid % q4) * 4, r0 = tid / q4;
        const int hq4 = 64, hc4 = (tid % hq4) * 4, hr0 = tid / hq4;  // H = 256: 4 rows per pass, 8 passes
        float4 av[npass], wv[npass], wh[NH][8];
#pragma unroll
        for (int q = 0; q < npass; ++q) {
            const int r = r0 + q * rpp;
            av[q] = *reinterpret_cast<const float4 *>(g.a0 + (size_t)min(r, R - 1) * C0 + c4);
            wv[q] = *reinterpret_cast<const float4 *>(g.L[0].W + (size_t)(col0 + r) * C0 + c4);
        }
#pragma unroll
        for (int l = 1; l < NLT; ++l)
#pragma unroll
            for (int q = 0; q < 8; ++q) wh[l - 1][q] = *reinterpret_cast<const float4 *>(g.L[l].W + (size_t)(col0 + hr0 + q * 4) * H + hc4);
#pragma unroll
        for (int q = 0; q < npass; ++q) {
            const int r = r0 + q * rpp;
            float4 a = av[q];
            const float ma = r < R ? 1.f : 0.f;
            a.x *= ma, a.y *= ma, a.z *= ma, a.w *= ma;
            *reinterpret_cast<float4 *>(As + r * LDA + c4) = a;
            *reinterpret_cast<float4 *>(W0s + r * (C0 + 4) + c4) = wv[q];
        }
#pragma unroll
        for (int l = 1; l < NLT; ++l)
#pragma unroll
            for (int q = 0; q < 8; ++q)
                *reinterpret_cast<float4 *>(Whs + (size_t)(l - 1) * 32 * (H + 4) + (hr0 + q * 4) * (H + 4) + hc4) = wh[l - 1][q];
    } else
    {
        const int q4 = C0 / 4, rpp = 256 / q4, npass = 32 / rpp;
        const int c4 = (tid % q4) * 4, r0 = tid / q4;
        for (int q = 0; q < npass; ++q) {
            const int r = r0 + q * rpp;
            float4 a = *reinterpret_cast<const float4 *>(g.a0 + (size_t)min(r, R - 1) * C0 + c4);
            const float ma = r < R ? 1.f : 0.f;
            a.x *= ma, a.y *= ma, a.z *= ma, a.w *= ma;
            *reinterpret_cast<float4 *>(As + r * LDA + c4) = a;
            *reinterpret_cast<float4 *>(W0s + r * (C0 + 4) + c4) =
                *reinterpret_cast<const float4 *>(g.L[0].W + (size_t)(col0 + r) * C0 + c4);
        }
        const int hq4 = H / 4, hrpp = 256 / hq4, hnpass = 32 / hrpp;
        const int hc4 = (tid % hq4) * 4, hr0 = tid / hq4;
        for (int l = 1; l < nl; ++l)
            for (int q = 0; q < hnpass; ++q) {
                const int r = hr0 + q * hrpp;
                *reinterpret_cast<float4 *>(Whs + (size_t)(l - 1) * 32 * (H + 4) + r * (H + 4) + hc4) =
                    *reinterpret_cast<const float4 *>(g.L[l].W + (size_t)(col0 + r) * H + hc4);
            }
    }
    lds_barrier();
    const unsigned epoch = s_epoch;
    FC_TL(0, wg, 1);

    for (int l = 0; l < nl; ++l) {
        const FcChainLayer &Lr = g.L[l];
        const int K = l == 0 ? C0 : H, LDW = K + 4;
        // epilogue inputs first (their latency hides under the MFMAs): this lane's epilogue column, see wave_reduce_scatter4
        const int er0 = 4 * (lane & 7), ecl = wave * 8 + (lane >> 3), ecol = col0 + ecl;
        const float ebias = Lr.bias[ecol], eg = Lr.gamma[ecol], eb = Lr.beta[ecol];
        float bn_rm = 0.f, bn_rv = 0.f;
        if (Lr.running_mean) bn_rm = Lr.running_mean[ecol], bn_rv = Lr.running_var[ecol];
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
        const int kph = K / 8, kb = wave * (K / 4) + h * kph;
        const float *wsl = l == 0 ? W0s : Whs + (size_t)(l - 1) * 32 * (H + 4);
        const float *ap = As + l31 * LDA + kb, *bp = wsl + l31 * LDW + kb;
        for (int t = 0; t < kph; t += 4) {
            const float4 a = *reinterpret_cast<const float4 *>(ap + t), b = *reinterpret_cast<const float4 *>(bp + t);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
        }
        const float4 zs4 = wave_reduce_scatter4(acc, red);
        FC_TL(0, wg, 2 + 6 * l);
        {
            // this lane: column ecol, rows er0 .. er0 + 3
            const float zv[4] = {zs4.x + ebias, zs4.y + ebias, zs4.z + ebias, zs4.w + ebias};
            float t[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) t[i] = er0 + i < R ? zv[i] : 0.f;
            const float s0 = col_sum_seq(t, lane & 7);
            const float meanf = s0 / (float)R;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float d = zv[i] - meanf;
                t[i] = er0 + i < R ? d * d : 0.f;
            }
            const float s2 = col_sum_seq(t, lane & 7);
            // (reciprocals from the host, Newton-refined reciprocal square root: three double divisions and a double square root
            //  per layer cost ~0.7 us of the chain's critical path)
            const double mean = (double)s0 * g.rinv_rows;
            const double dm = mean - (double)meanf;
            double var = (double)s2 * g.rinv_rows - dm * dm;
            if (var < 0.0) var = 0.0;
            const float invstd = (float)fast_rsqrt(var + (double)Lr.eps);
            const float sc = eg * invstd, sh = eb - (float)mean * sc;
            if ((lane & 7) == 0) {
                Lr.coef[ecol] = sc, Lr.coef[H + ecol] = sh, Lr.coef[2 * H + ecol] = (float)mean, Lr.coef[3 * H + ecol] = invstd;
                if (Lr.running_mean) {
                    const double unbiased = var * g.unbias;
                    Lr.running_mean[ecol] = (1.f - Lr.momentum) * bn_rm + Lr.momentum * (float)mean;
                    Lr.running_var[ecol] = (1.f - Lr.momentum) * bn_rv + Lr.momentum * (float)unbiased;
                }
                if (wg == 0 && tid == 0 && Lr.num_batches_tracked) *Lr.num_batches_tracked += 1;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = er0 + i;
                Ts[row * 36 + ecl] = zv[i];
                Ta[row * 36 + ecl] = row < R ? relu_np(fmaf(zv[i], sc, sh)) : 0.f;
            }
        }
        lds_barrier();
        FC_TL(0, wg, 3 + 6 * l);
        // the 32 x 32 tiles leave as 16-byte stores: thread -> (row = tid / 8, 4 columns at (tid % 8) * 4)
        const int trow = tid >> 3, tc4 = (tid & 7) * 4;
        if (trow < R) {
            float4 zv = *reinterpret_cast<const float4 *>(Ts + trow * 36 + tc4);
            // a seam of this launch timed out: what the layers above computed is built on incomplete activations -- the head's
            // output must not look like a result (NaN flows through fc4 / the pair scan into the loss and every gradient)
            if (!OUT && l == nl - 1 && s_bad) zv.x = zv.y = zv.z = zv.w = __builtin_nanf("");
            *reinterpret_cast<float4 *>(Lr.z + (size_t)trow * H + col0 + tc4) = zv;
        }
        if (!OUT && l == nl - 1) break;
        float *xb = g.xbuf + (size_t)(l & 1) * 32 * H;
        {
            const float4 v = *reinterpret_cast<const float4 *>(Ta + trow * 36 + tc4);
            f32x4v vv = {v.x, v.y, v.z, v.w};
            store_sc1_b128(xb + (size_t)trow * H + col0 + tc4, vv);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        lds_barrier();
        FC_TL(0, wg, 4 + 6 * l);
        if (tid == 0) {
            if (fc_chain_seam(g.sync, 1 + l, epoch, nwg, 1u + (unsigned)l, (int)s_limit)) s_bad = 1;
            if (l == 0 && wg == 0) __hip_atomic_store(g.sync, epoch + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        lds_barrier();
        FC_TL(0, wg, 5 + 6 * l);
        if (POOL && l == 0 && tid < 16) {  // every workgroup is past the pool stage: this one's 16 channels of the sums can go
            const int c = wg * 16 + tid;
#pragma unroll
            for (int q = 0; q < kFxSlots * 2 + 2; ++q) g.P.acc[q * kFxRow + c] = 0;  // lo rows and the two hi rows (the poison
        }                                                                           // word: first kernel of the next step)
        // gather the whole 32 x H activation (write-through data: sc1 loads read it from L2 / memory, never from a stale L1 line)
        {
            const int hq4 = H / 4;
            f32x4v r[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int idx = tid + 256 * i;
                const float *p = xb + (size_t)(idx / hq4) * H + (idx % hq4) * 4;
                asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=&v"(r[i]) : "v"(p) : "memory");
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            FC_TL(0, wg, 6 + 6 * l);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int idx = tid + 256 * i;
                if (idx < 32 * hq4)
                    *reinterpret_cast<float4 *>(As + (idx / hq4) * LDA + (idx % hq4) * 4) = make_float4(r[i].x, r[i].y, r[i].z, r[i].w);
            }
        }
        if constexpr (OUT) {
            if (l == 1) {  // layer 1's MFMAs are long done (several barriers ago): its slice's slot takes the output layer's
                const int hc4 = (tid % 64) * 4, hr0 = tid / 64;
#pragma unroll
                for (int q = 0; q < 8; ++q) *reinterpret_cast<f32x4v *>(Whs + (hr0 + q * 4) * (H + 4) + hc4) = wo[q];
            }
        }
        lds_barrier();
        FC_TL(0, wg, 7 + 6 * l);
    }
    if constexpr (OUT) {
        // ---- the output layer: y = bn(As W_out^T + b), no activation.  As holds the last hidden layer's activations (gathered above)
        const FcChainOut &O = g.O;
        const int Co = O.Co;
        if (col0 < Co) {
            const int er0 = 4 * (lane & 7), ecl = wave * 8 + (lane >> 3), ecol = col0 + ecl;
            const float ebias = O.bias[ecol], eg = O.gamma[ecol], eb = O.beta[ecol];
            float bn_rm = 0.f, bn_rv = 0.f;
            if (O.running_mean) bn_rm = O.running_mean[ecol], bn_rv = O.running_var[ecol];
            f32x16 acc;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = 0.f;
            const int kph = H / 8, kb = wave * (H / 4) + h * kph;
            const float *ap = As + l31 * LDA + kb, *bp = Whs + l31 * (H + 4) + kb;
            for (int t = 0; t < kph; t += 4) {
                const float4 a = *reinterpret_cast<const float4 *>(ap + t), b = *reinterpret_cast<const float4 *>(bp + t);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
            }
            const float4 zs4 = wave_reduce_scatter4(acc, red);
            {
                const float zv[4] = {zs4.x + ebias, zs4.y + ebias, zs4.z + ebias, zs4.w + ebias};
                float t[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) t[i] = er0 + i < R ? zv[i] : 0.f;
                const float s0 = col_sum_seq(t, lane & 7);
                const float meanf = s0 / (float)R;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float d = zv[i] - meanf;
                    t[i] = er0 + i < R ? d * d : 0.f;
                }
                const float s2 = col_sum_seq(t, lane & 7);
                const double mean = (double)s0 * g.rinv_rows;
                const double dm = mean - (double)meanf;
                double var = (double)s2 * g.rinv_rows - dm * dm;
                if (var < 0.0) var = 0.0;
                const float invstd = (float)fast_rsqrt(var + (double)O.eps);
                const float sc = eg * invstd, sh = eb - (float)mean * sc;
                if ((lane & 7) == 0) {
                    O.coef[ecol] = sc, O.coef[Co + ecol] = sh, O.coef[2 * Co + ecol] = (float)mean, O.coef[3 * Co + ecol] = invstd;
                    if (O.running_mean) {
                        const double unbiased = var * g.unbias;
                        O.running_mean[ecol] = (1.f - O.momentum) * bn_rm + O.momentum * (float)mean;
                        O.running_var[ecol] = (1.f - O.momentum) * bn_rv + O.momentum * (float)unbiased;
                    }
                    if (wg == 0 && tid == 0 && O.num_batches_tracked) *O.num_batches_tracked += 1;
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int row = er0 + i;
                    Ts[row * 36 + ecl] = zv[i];
                    Ta[row * 36 + ecl] = fmaf(zv[i], sc, sh);
                }
            }
            lds_barrier();
            const int trow = tid >> 3, tc4 = (tid & 7) * 4;
            if (trow < R) {
                float4 zv = *reinterpret_cast<const float4 *>(Ts + trow * 36 + tc4), yv = *reinterpret_cast<const float4 *>(Ta + trow * 36 + tc4);
                // a seam of this launch timed out: the head's output must not look like a result (see the hidden layers' note)
                if (s_bad) yv.x = yv.y = yv.z = yv.w = zv.x = zv.y = zv.z = zv.w = __builtin_nanf("");
                *reinterpret_cast<float4 *>(O.z + (size_t)trow * Co + col0 + tc4) = zv;
                *reinterpret_cast<float4 *>(O.y + (size_t)trow * Co + col0 + tc4) = yv;
            }
        }
    }
    SN_TL_DRAIN();
    FC_TL(0, wg, 31);
}

// ------------------------------------------------------------------------------------------------
// FC head backward as ONE launch (rows R <= 32, every width <= 256): the mirror of fc_chain_fwd_kernel.  Stage s is the
// backward of GEMM layer l (top layer first).  16 resident workgroups on one XCD in two roles:
//   * 8 DATA-GRADIENT workgroups form the dependency chain: workgroup j owns columns [32 j, 32 j + 32) -- dZ_l (32 x Co, in
//     LDS) times the column slice of W_l (staged TRANSPOSED in LDS, fetched one stage ahead), ReLU mask and BatchNorm backward
//     of the layer below in wave 0's epilogue (statistics are per column over the rows: local) -> its 32 x 32 tile of
//     dZ_{l-1} leaves as write-through stores into the stage's own slab, drained, one arrival atomic; poll; gather dZ_{l-1}.
//   * 8 WEIGHT-GRADIENT workgroups hang off that chain without being on it: workgroup j waits for the stage's arrivals, takes
//     tile j of dZ_l (stage 0: straight from the incoming gradient) and computes rows [32 j, 32 j + 32) of
//     dW_l = dZ_l^T . A_{l-1} (operands of A_{l-1} prefetched from the forward's tensors), one wave per 32 x 32 tile.
// Splitting the roles matters because ONE CU moves only ~50-100 GB/s: with both jobs on the same 8 CUs every stage pulled
// ~160 KB through one CU and the launch was no faster than the four it replaces (measured 36 us); the chain workgroups now
// move ~70 KB per stage.  Nothing but the parameter gradients, the pooled-feature gradient (gsel) and the top conv
// BatchNorm's dZ coefficients goes back to HBM as tensors.  Synchronisation: one slab and one monotonic arrival counter per
// stage (no reuse inside a launch); the launch epoch (sync[0]) is read by all 16 workgroups, each confirms on sync[14], and
// chain workgroup 0 advances it at its end once all 16 confirmations are in.  Arithmetic (K split over the waves, MFMA order,
// epilogue expressions) is that of small_dgrad_body / small_wgrad_body.
// ------------------------------------------------------------------------------------------------
constexpr int kFcBwdMaxStages = 5;
struct FcBwdStage {
    const float *W;  // (Co, Ci) of the layer this stage differentiates
    int Co, Ci;
    const float *zprev, *coefprev;  // layer below: pre-BN output (R, Ci) at this layer's inputs and its (4, Ci) coefficients
    long long bn_rows;              // rows the BatchNorm below averaged over (R, B * N behind the max-pool, < 0: fixed statistics)
    double rinv;                    // 1 / bn_rows (0 for fixed statistics): from the host, a double division costs ~0.2 us here
    float *dgamma, *dbeta, *dbias;  // of the layer below
    float *dW, *db;                 // of this layer (db: top layer only)
    const float *aprev;             // wgrad operand: zprev (relu(bn(.)) applied) or, araw != 0, the raw input (pooled features)
    int araw;
    float *gout, *kout;  // optional (last stage): masked gradient (R, Ci) and kcoef (3, Ci) of the layer below to HBM
};
struct FcBwdArgs {
    const float *gy;  // (R, Co of stage 0): gradient w.r.t. the head's output
    // Optional: the head's output went through a BatchNorm WITHOUT activation (classification sampler, FcChainOut): gy is the
    // gradient behind it, and stage 0 opens with that BatchNorm's backward -- oz (R, Co) its input, ocoef (4, Co) the forward's
    // coefficients -> dZ = k1 gy + k2 z + k3 per column (sums over the R rows: local to every workgroup that holds the columns),
    // odgamma / odbeta (Co) written by chain workgroup 0.  ofixed: the forward ran on running statistics (dZ = scale gy).
    const float *oz, *ocoef;
    float *odgamma, *odbeta;
    int ofixed;
    int R, ns;
    FcBwdStage S[kFcBwdMaxStages];
    float *xbuf;     // [ns][32][256] hand-off slabs, one per stage
    unsigned *sync;  // [0] epoch, [1 + s] arrivals of stage s, [14] epoch readers, [15] error flag -- persistent, zeroed once
};

__device__ __forceinline__ bool fc_wait_arrivals(unsigned *ctr, unsigned target, int limit)
{
    int spins = 0;
    while ((int)(__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
        if (++spins > limit) return false;  // never on a healthy run: report instead of hanging the device (fc_chain_seam)
        __builtin_amdgcn_s_sleep(1);
    }
    return true;
}

// Backward of the output BatchNorm on `ncols` columns held in LDS: G [32][ldg] the upstream gradient (rewritten in place with
// dZ), Z [32][ldz] the BatchNorm's input; thread t < ncols owns column t (global column c0 + t).  Sums in double, rows in order.
__device__ __forceinline__ void fc_out_bn_backward(const FcBwdArgs &g, float *G, int ldg, const float *Z, int ldz, int ncols, int c0, int Co,
                                                   bool write_params)
{
    const int t = threadIdx.x;
    if (t < ncols) {
        const int c = c0 + t;
        const float scale = g.ocoef[c], mean = g.ocoef[2 * Co + c], invstd = g.ocoef[3 * Co + c];
        double s0 = 0.0, s1 = 0.0;
        for (int r = 0; r < g.R; ++r) {
            const float gv = G[r * ldg + t], zv = Z[r * ldz + t];
            s0 += (double)gv;
            s1 += (double)gv * (double)(zv - mean);
        }
        const double dg = (double)invstd * s1, rinv = 1.0 / (double)g.R, sc = scale;
        const float k1 = scale;
        const float k2 = g.ofixed ? 0.f : (float)(-sc * (double)invstd * dg * rinv);
        const float k3 = g.ofixed ? 0.f : (float)(sc * ((double)invstd * (double)mean * dg * rinv - s0 * rinv));
        if (write_params) g.odgamma[c] = (float)dg, g.odbeta[c] = (float)s0;
        for (int r = 0; r < g.R; ++r) G[r * ldg + t] = fmaf(k1, G[r * ldg + t], fmaf(k2, Z[r * ldz + t], k3));
    }
}

__global__ void __launch_bounds__(256) fc_chain_bwd_kernel(FcBwdArgs g)
{
    extern __shared__ __attribute__((aligned(16))) float sm[];
    __shared__ unsigned s_epoch, s_limit, s_bad;
    if (blockIdx.x & 7) return;
    const int wgi = blockIdx.x >> 3;  // 0..7 data-gradient chain, 8..15 weight gradients
    constexpr int NWG = 8, LD = 256 + 4;
    const bool chain = wgi < NWG;
    const int wg = chain ? wgi : wgi - NWG;
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int R = g.R, ns = g.ns;
    const int col0 = wg * 32;
    if (tid == 0) {
        const unsigned lim = g.sync[13 * kFcSyncStride];  // poll bound override (tests)
        s_epoch = __hip_atomic_load(g.sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(g.sync + 14 * kFcSyncStride, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_limit = lim ? lim : (unsigned)kFcChainPolls, s_bad = 0;
    }
    lds_barrier();
    FC_TL(1, wgi, 0);
    const unsigned epoch = s_epoch;
    const unsigned target = (epoch + 1u) * (unsigned)NWG;
    const int limit = (int)s_limit;
    const float kNaN = __builtin_nanf("");

    if (!chain) {
        // ================================================================== weight-gradient workgroup
        float *Tz = sm;            // [32][36] tile wg of dZ_l
        float *Tw = sm + 32 * 36;  // [4][32][36] output tiles on their way out
        for (int s = 0; s < ns; ++s) {
            const FcBwdStage &S = g.S[s];
            const int Co = S.Co, Ci = S.Ci, tiles_n = Ci / 32;
            if (col0 >= Co) continue;  // (this layer has fewer than 32 (wg + 1) outputs)
            // operand tiles of A_{l-1} first: they do not depend on the chain
            float bw[2][16];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int tn = wave + 4 * i;
                if (tn < tiles_n) {
                    const int n = tn * 32 + l31;
                    float sc = 1.f, sh = 0.f;
                    if (!S.araw) sc = S.coefprev[n], sh = S.coefprev[Ci + n];
#pragma unroll
                    for (int t = 0; t < 16; ++t) {
                        const int r = h * 16 + t;
                        float v = S.aprev[(size_t)(r < R ? r : 0) * Ci + n];
                        if (!S.araw) v = relu_np(fmaf(v, sc, sh));
                        bw[i][t] = r < R ? v : 0.f;
                    }
                }
            }
            // tile wg of dZ_l: stage 0 from the incoming gradient, else from the slab the chain filled in stage s - 1
            const int trow = tid >> 3, tc4 = (tid & 7) * 4;
            if (s == 0) {
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (trow < R) v = *reinterpret_cast<const float4 *>(g.gy + (size_t)trow * Co + col0 + tc4);
                *reinterpret_cast<float4 *>(Tz + trow * 36 + tc4) = v;
                if (g.oz) {  // the output BatchNorm's backward on this workgroup's 32 columns (its input tile parked in Tw)
                    float4 zv = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (trow < R) zv = *reinterpret_cast<const float4 *>(g.oz + (size_t)trow * Co + col0 + tc4);
                    *reinterpret_cast<float4 *>(Tw + trow * 36 + tc4) = zv;
                    lds_barrier();
                    fc_out_bn_backward(g, Tz, 36, Tw, 36, 32, col0, Co, false);
                }
            } else {
                if (tid == 0 && !fc_wait_arrivals(g.sync + s * kFcSyncStride, target, limit)) {  // sync[1 + (s - 1)]
                    __hip_atomic_store(g.sync + 15 * kFcSyncStride, 16u + (unsigned)s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    s_bad = 1;
                }
                lds_barrier();
                const float *p = g.xbuf + (size_t)(s - 1) * 32 * 256 + (size_t)trow * Co + col0 + tc4;
                f32x4v v;
                asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
                *reinterpret_cast<float4 *>(Tz + trow * 36 + tc4) = make_float4(v.x, v.y, v.z, v.w);
            }
            lds_barrier();
            float a[16];
#pragma unroll
            for (int t = 0; t < 16; ++t) a[t] = Tz[(h * 16 + t) * 36 + l31];
            float *T = Tw + wave * 32 * 36;
            const bool bad = s_bad != 0;  // a hand-off this workgroup waited for never came: its gradients must not look like results
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int tn = wave + 4 * i;
                if (tn < tiles_n) {
                    f32x16 acc;
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
                    for (int t = 0; t < 16; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], bw[i][t], acc, 0, 0, 0);
#pragma unroll
                    for (int e = 0; e < 16; ++e) T[frag_row(e, lane) * 36 + l31] = bad ? kNaN : acc[e];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int rt = 8 * q + (lane >> 3);
                        *reinterpret_cast<float4 *>(S.dW + (size_t)(col0 + rt) * Ci + tn * 32 + (lane & 7) * 4) =
                            *reinterpret_cast<const float4 *>(T + rt * 36 + (lane & 7) * 4);
                    }
                }
            }
            if (S.db && wave == 0) {  // bias gradient of the top layer: the ones-column MFMA of small_wgrad_body
                f32x16 acc;
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
                for (int t = 0; t < 16; ++t)
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], (l31 == 0 && h * 16 + t < R) ? 1.f : 0.f, acc, 0, 0, 0);
                if (l31 == 0)
#pragma unroll
                    for (int e = 0; e < 16; ++e) S.db[col0 + frag_row(e, lane)] = acc[e];
            }
            lds_barrier();  // Tz is rewritten by the next stage
            FC_TL(1, wgi, 2 + 6 * s);
        }
        SN_TL_DRAIN();
        FC_TL(1, wgi, 31);
        return;
    }

    // ====================================================================== data-gradient chain workgroup
    float *dZs = sm;                 // [32][LD]   dZ of the current layer, all columns
    float *Wt0 = dZs + 32 * LD;      // [2][32][LD] transposed weight slices (ci within the tile, co)
    float *red = Wt0 + 2 * 32 * LD;  // [4][64][kRsPitch]: the waves' K partials (wave_reduce_scatter4)
    float *Ta = red + kRsFloats;     // [32][36]   this workgroup's tile of dZ of the layer below
    // ---- stage 0 operands: dZ of the top layer straight from HBM, its weight slice (transposed).  ALL loads are issued before
    // the first LDS write, unconditionally (out-of-range slots re-read a valid address): a loop of load -> store iterations, or
    // a load behind a branch, makes every iteration pay its own memory round trip (measured: 4.2 us for this block before)
    {
        const FcBwdStage &S = g.S[0];
        const int q4 = S.Co / 4, ng = 32 * q4, nw = S.Co * 8;  // float4 per row; float4 of the gradient / of the weight slice
        const bool wt = col0 < S.Ci;
        float4 gv[8], wv[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int idx = min(tid + 256 * i, ng - 1), r = idx / q4;
            gv[i] = *reinterpret_cast<const float4 *>(g.gy + (size_t)min(r, R - 1) * S.Co + (idx % q4) * 4);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int idx = min(tid + 256 * i, nw - 1);
            wv[i] = *reinterpret_cast<const float4 *>(S.W + (size_t)(idx >> 3) * S.Ci + (wt ? col0 : 0) + (idx & 7) * 4);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int idx = tid + 256 * i;
            if (idx < ng) {
                const int r = idx / q4;
                *reinterpret_cast<float4 *>(dZs + r * LD + (idx % q4) * 4) = r < R ? gv[i] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        if (wt)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int idx = tid + 256 * i;
                if (idx < nw) {
                    const int co = idx >> 3, c4 = (idx & 7) * 4;
                    Wt0[(c4 + 0) * LD + co] = wv[i].x, Wt0[(c4 + 1) * LD + co] = wv[i].y, Wt0[(c4 + 2) * LD + co] = wv[i].z, Wt0[(c4 + 3) * LD + co] = wv[i].w;
                }
            }
        if (g.oz) {
            // the head's output went through a BatchNorm: gy -> dZ of the output layer, all Co columns, before the first product.
            // Its input z parks in the weight buffer of stage 1 (free until the first hand-off).
            float *Zo = Wt0 + 32 * LD;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int idx = tid + 256 * i;
                if (idx < ng) {
                    const int r = idx / q4;
                    float4 zv = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (r < R) zv = *reinterpret_cast<const float4 *>(g.oz + (size_t)r * S.Co + (idx % q4) * 4);
                    *reinterpret_cast<float4 *>(Zo + r * LD + (idx % q4) * 4) = zv;
                }
            }
            lds_barrier();
            fc_out_bn_backward(g, dZs, LD, Zo, LD, S.Co, 0, S.Co, wgi == 0);
        }
    }
    lds_barrier();
    FC_TL(1, wgi, 1);

    for (int s = 0; s < ns; ++s) {
        const FcBwdStage &S = g.S[s];
        const int Co = S.Co, Ci = S.Ci;
        const bool has_tile = col0 < Ci, more = s + 1 < ns;
        float *Wt = Wt0 + (size_t)(s & 1) * 32 * LD, *Wtn = Wt0 + (size_t)((s + 1) & 1) * 32 * LD;
        // ---- prefetches: next stage's weight slice (all four waves fetch and later stage it); the epilogue's inputs
        constexpr int NWN = 8;  // 256 rows x 8 float4 / 256 threads
        float4 wn[NWN];
        bool next_tile = false;
        int nCo = 0, nCi = 0;
        if (more) {
            const FcBwdStage &N = g.S[s + 1];
            nCo = N.Co, nCi = N.Ci;
            next_tile = col0 < nCi;
            if (next_tile)
#pragma unroll
                for (int i = 0; i < NWN; ++i) {  // unconditional (clamped): a load behind a branch is waited for on the spot
                    const int idx = min(tid + 256 * i, nCo * 8 - 1);
                    wn[i] = *reinterpret_cast<const float4 *>(N.W + (size_t)(idx >> 3) * nCi + col0 + (idx & 7) * 4);
                }
        }
        // the epilogue runs on all four waves (wave_reduce_scatter4): this lane -> column ecol, rows er0 .. er0 + 3
        const int er0 = 4 * (lane & 7), ecl = wave * 8 + (lane >> 3), ecol = col0 + ecl;
        float zpv[4], esc = 0.f, esh = 0.f, pmean = 0.f, pinv = 0.f;
        if (has_tile) {
            esc = S.coefprev[ecol], esh = S.coefprev[Ci + ecol], pmean = S.coefprev[2 * Ci + ecol], pinv = S.coefprev[3 * Ci + ecol];
#pragma unroll
            for (int i = 0; i < 4; ++i) zpv[i] = S.zprev[er0 + i < R ? (size_t)(er0 + i) * Ci + ecol : 0];
        }
        // ---- data gradient tile
        if (has_tile) {
            f32x16 acc;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = 0.f;
            for (int k0 = wave * 2 * KP; k0 < Co; k0 += 4 * 2 * KP) {
                const int kb = k0 + h * KP;
                const float *ap = dZs + l31 * LD + kb, *bp = Wt + l31 * LD + kb;
#pragma unroll
                for (int t = 0; t < KP; t += 4) {
                    const float4 a = *reinterpret_cast<const float4 *>(ap + t), b = *reinterpret_cast<const float4 *>(bp + t);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
                }
            }
            const float4 dz4 = wave_reduce_scatter4(acc, red);
            FC_TL(1, wgi, 2 + 6 * s);
            {
                const float dv[4] = {dz4.x, dz4.y, dz4.z, dz4.w};
                float gv[4], t0[4], t1[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const bool live = er0 + i < R;
                    gv[i] = (live && fmaf(zpv[i], esc, esh) > 0.f) ? dv[i] : 0.f;
                    t0[i] = gv[i];
                    t1[i] = live ? gv[i] * (zpv[i] - pmean) : 0.f;
                }
                const float s0 = col_sum_seq(t0, lane & 7), s1c = col_sum_seq(t1, lane & 7);
                const double scale = esc, mean = pmean, invstd = pinv, sd = s0;
                const double dg = invstd * (double)s1c;
                const double rinv = S.rinv;
                const float k1 = (float)scale, k2 = (float)(-scale * invstd * dg * rinv);
                const float k3 = (float)(scale * (invstd * mean * dg * rinv - sd * rinv));
                if ((lane & 7) == 0) {
                    S.dgamma[ecol] = (float)dg, S.dbeta[ecol] = (float)sd;
                    if (S.dbias)
                        S.dbias[ecol] = (float)((double)k1 * sd + (double)k2 * (double)S.bn_rows * mean + (double)S.bn_rows * (double)k3);
                    if (S.kout) S.kout[ecol] = k1, S.kout[Ci + ecol] = k2, S.kout[2 * Ci + ecol] = k3;
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int row = er0 + i;
                    if (S.gout && row < R) S.gout[(size_t)row * Ci + ecol] = s_bad ? kNaN : gv[i];  // (a seam timed out: see fc_chain_seam)
                    Ta[row * 36 + ecl] = row < R ? fmaf(k1, gv[i], fmaf(k2, zpv[i], k3)) : 0.f;
                }
            }
        }
        lds_barrier();
        FC_TL(1, wgi, 3 + 6 * s);
        if (!more) break;
        // ---- hand the tile of dZ of the layer below over (write-through), drained, then arrive
        float *xb = g.xbuf + (size_t)s * 32 * 256;
        if (has_tile) {
            const int trow = tid >> 3, tc4 = (tid & 7) * 4;
            const float4 v = *reinterpret_cast<const float4 *>(Ta + trow * 36 + tc4);
            f32x4v vv = {v.x, v.y, v.z, v.w};
            store_sc1_b128(xb + (size_t)trow * Ci + col0 + tc4, vv);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        lds_barrier();
        FC_TL(1, wgi, 4 + 6 * s);
        if (tid == 0) __hip_atomic_fetch_add(g.sync + (1 + s) * kFcSyncStride, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // ---- the next stage's weight slice goes into the other LDS buffer while the other workgroups' arrivals are awaited
        // (the poll below cannot succeed sooner than the slowest of them: the staging is free there; in front of the hand-off
        // it delayed this workgroup's own arrival)
        if (next_tile)
#pragma unroll
            for (int i = 0; i < NWN; ++i) {
                const int idx = tid + 256 * i;
                if (idx < nCo * 8) {
                    const int co = idx >> 3, c4 = (idx & 7) * 4;
                    Wtn[(c4 + 0) * LD + co] = wn[i].x, Wtn[(c4 + 1) * LD + co] = wn[i].y;
                    Wtn[(c4 + 2) * LD + co] = wn[i].z, Wtn[(c4 + 3) * LD + co] = wn[i].w;
                }
            }
        if (tid == 0 && !fc_wait_arrivals(g.sync + (1 + s) * kFcSyncStride, target, limit)) {
            __hip_atomic_store(g.sync + 15 * kFcSyncStride, 1u + (unsigned)s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_bad = 1;
        }
        lds_barrier();
        FC_TL(1, wgi, 5 + 6 * s);
        {   // gather dZ of the layer below (32 x Ci; Ci = 256: 8 loads per thread, 128: 4)
            const int q4 = Ci / 4, nld = (32 * q4) / 256;
            f32x4v r[8];
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (i < nld) {
                    const int idx = tid + 256 * i;
                    const float *p = xb + (size_t)(idx / q4) * Ci + (idx % q4) * 4;
                    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=&v"(r[i]) : "v"(p) : "memory");
                }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            FC_TL(1, wgi, 6 + 6 * s);
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (i < nld) {
                    const int idx = tid + 256 * i;
                    *reinterpret_cast<float4 *>(dZs + (idx / q4) * LD + (idx % q4) * 4) = make_float4(r[i].x, r[i].y, r[i].z, r[i].w);
                }
        }
        lds_barrier();
        FC_TL(1, wgi, 7 + 6 * s);
    }
    SN_TL_DRAIN();
    FC_TL(1, wgi, 31);
    // the next launch may only see the advanced epoch once all 16 workgroups of this one have read the current value
    if (wgi == 0 && tid == 0) {
        if (!fc_wait_arrivals(g.sync + 14 * kFcSyncStride, (epoch + 1u) * 16u, limit))
            __hip_atomic_store(g.sync + 15 * kFcSyncStride, 64u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(g.sync, epoch + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

}  // namespace sn

using namespace sn;

static size_t fc_chain_fwd_lds(int C0, int H, int nl)
{
    const int LDA = (C0 > H ? C0 : H) + 4;
    return ((size_t)32 * LDA + (size_t)32 * (C0 + 4) + (size_t)(nl - 1) * 32 * (H + 4) + kRsFloats + 2 * 32 * 36) * sizeof(float);
}

// 1: sn_fc_chain_forward runs this FC head (R rows, C0 -> H -> ... -> H, nl BatchNorm + ReLU layers) as one launch
extern "C" int sn_fc_chain_forward_supported(int R, int C0, int H, int nl)
{
    return R >= 1 && R <= 32 && nl >= 2 && nl <= kFcChainMaxLayers && H == 256 && (C0 == 64 || C0 == 128 || C0 == 256) &&
           fc_chain_fwd_lds(C0, H, nl) <= (size_t)160 * 1024 - 64;
}

extern "C" int sn_fc_chain_forward(int R, int C0, int H, int nl, const float *a0, const float *const *W, const float *const *bias,
                                   const float *const *gamma, const float *const *beta, float *const *running_mean,
                                   float *const *running_var, long long *const *num_batches_tracked, const float *eps,
                                   const float *momentum, float *const *z, float *const *coef, float *xbuf, unsigned *sync,
                                   sn_stream_t stream)
{
    SN_REQUIRE(sn_fc_chain_forward_supported(R, C0, H, nl), "shape not supported (sn_fc_chain_forward_supported)");
    SN_REQUIRE(a0 && W && bias && gamma && beta && eps && momentum && z && coef && xbuf && sync, "null pointer");
    FcChainArgs g{};
    g.a0 = a0, g.R = R, g.C0 = C0, g.H = H, g.nl = nl, g.xbuf = xbuf, g.sync = sync;
    for (int l = 0; l < nl; ++l) {
        SN_REQUIRE(W[l] && bias[l] && gamma[l] && beta[l] && z[l] && coef[l], "null layer pointer");
        g.L[l] = FcChainLayer{W[l], bias[l], gamma[l], beta[l], running_mean ? running_mean[l] : nullptr,
                              running_var ? running_var[l] : nullptr, num_batches_tracked ? num_batches_tracked[l] : nullptr,
                              z[l], coef[l], eps[l], momentum[l]};
    }
    const size_t lds = fc_chain_fwd_lds(C0, H, nl);
    {   // (the LDS size depends on the shape: renewed per device whenever a call needs more than was granted)
        static SnLdsAttrGrow a0, a1, a2;
        if (sn_lds_attr_grow(a0, (const void *)fc_chain_fwd_kernel<128, 3>, lds, "sn_fc_chain_forward") ||
            sn_lds_attr_grow(a1, (const void *)fc_chain_fwd_kernel<256, 3>, lds, "sn_fc_chain_forward") ||
            sn_lds_attr_grow(a2, (const void *)fc_chain_fwd_kernel<0, 0>, lds, "sn_fc_chain_forward"))
            return SN_ERR_UNSUPPORTED;
    }
    // 8 x (H / 32) blocks: block b lands on XCD b % 8, the b % 8 == 0 ones do the work -- all on one XCD (same L2)
    g.rinv_rows = 1.0 / (double)R, g.unbias = R > 1 ? (double)R / (double)(R - 1) : 1.0;
    const dim3 grid(8 * (H / 32)), block(256);
    if (C0 == 128 && nl == 3)
        hipLaunchKernelGGL((fc_chain_fwd_kernel<128, 3>), grid, block, lds, (hipStream_t)stream, g);
    else if (C0 == 256 && nl == 3)
        hipLaunchKernelGGL((fc_chain_fwd_kernel<256, 3>), grid, block, lds, (hipStream_t)stream, g);
    else
        hipLaunchKernelGGL((fc_chain_fwd_kernel<0, 0>), grid, block, lds, (hipStream_t)stream, g);
    SN_LAUNCH_CHECK();
    return 0;
}

// sn_fc_chain_forward with the tail of the conv stack in front: sn_conv_stack_forward_bn called with pooled = argsel = zsel =
// NULL stops after its last GEMM; this launch then finalises that layer's BatchNorm from the fixed-point sums in acc (clearing
// them), picks the max-pool from pool_val / pool_idx and runs the FC head on the result -- one launch and one ~2 us seam
// instead of a 6 us kernel and its boundary.  B <= 32 clouds of N <= 1024 points (N % 64 == 0), nconv conv layers ending in
// 128 channels, FC head 128 -> 256 x 3.  gamma5 .. coef5: the last conv layer's BatchNorm as in sn_conv_stack_forward_bn.
extern "C" int sn_fc_chain_forward_pool_supported(int B, int N, int C0, int H, int nl)
{
    return B >= 1 && B <= 32 && N >= 64 && N % 64 == 0 && C0 == 128 && H == 256 && nl == 3;
}

extern "C" int sn_fc_chain_forward_pool(int B, int N, int nconv, long long *acc, const float *pool_val, const int *pool_idx,
                                        const float *gamma5, const float *beta5, float *running_mean5, float *running_var5,
                                        long long *num_batches_tracked5, float eps5, float momentum5, float *coef5, float *pooled,
                                        int *argsel, float *zsel, int H, int nl, const float *const *W, const float *const *bias,
                                        const float *const *gamma, const float *const *beta, float *const *running_mean,
                                        float *const *running_var, long long *const *num_batches_tracked, const float *eps,
                                        const float *momentum, float *const *z, float *const *coef, float *xbuf, unsigned *sync,
                                        sn_stream_t stream)
{
    constexpr int C0 = 128;
    SN_REQUIRE(sn_fc_chain_forward_pool_supported(B, N, C0, H, nl) && nconv >= 2, "shape not supported (sn_fc_chain_forward_pool_supported)");
    SN_REQUIRE(acc && pool_val && pool_idx && gamma5 && beta5 && running_mean5 && running_var5 && coef5 && pooled && argsel && zsel,
               "null pointer");
    SN_REQUIRE(W && bias && gamma && beta && eps && momentum && z && coef && xbuf && sync, "null pointer");
    FcChainArgs g{};
    g.a0 = pooled, g.R = B, g.C0 = C0, g.H = H, g.nl = nl, g.xbuf = xbuf, g.sync = sync;
    g.P.acc = acc + (size_t)(nconv - 1) * kFxLayer, g.P.zero_ptr = acc + (size_t)(nconv - 2) * kFxLayer, g.P.zero_n = kFxLayer;
    g.P.keys = reinterpret_cast<const unsigned long long *>(pool_val);  // (B, 2, 128) keys: see sn_conv_stack_forward_bn, pooled == NULL
    (void)pool_idx;
    g.P.bn = BnFwd{gamma5, beta5, running_mean5, running_var5, num_batches_tracked5, coef5, eps5, momentum5, (long long)B * N};
    g.P.pooled = pooled, g.P.argsel = argsel, g.P.zsel = zsel;
    for (int l = 0; l < nl; ++l) {
        SN_REQUIRE(W[l] && bias[l] && gamma[l] && beta[l] && z[l] && coef[l], "null layer pointer");
        g.L[l] = FcChainLayer{W[l], bias[l], gamma[l], beta[l], running_mean ? running_mean[l] : nullptr,
                              running_var ? running_var[l] : nullptr, num_batches_tracked ? num_batches_tracked[l] : nullptr,
                              z[l], coef[l], eps[l], momentum[l]};
    }
    const size_t lds = fc_chain_fwd_lds(C0, H, nl);
    static SnLdsAttrGrow attr;
    if (sn_lds_attr_grow(attr, (const void *)fc_chain_fwd_kernel<128, 3, true>, lds, "sn_fc_chain_forward_pool")) return SN_ERR_UNSUPPORTED;
    g.rinv_rows = 1.0 / (double)B, g.unbias = B > 1 ? (double)B / (double)(B - 1) : 1.0;
    hipLaunchKernelGGL((fc_chain_fwd_kernel<128, 3, true>), dim3(8 * (H / 32)), dim3(256), lds, (hipStream_t)stream, g);
    SN_LAUNCH_CHECK();
    return 0;
}

// sn_fc_chain_forward_pool with the head's OUTPUT layer -- Linear (Co x H) + BatchNorm without activation (classification/models/
// samplenet_model.py:100-108) -- as the chain's last stage: Wo (Co, H), bo, gamma_o / beta_o (Co), running statistics (may be NULL),
// -> zo (B, Co) pre-BatchNorm, coef_o (4, Co), y (B, Co).  Co a multiple of 32, at most H.
extern "C" int sn_fc_chain_forward_pool_out_supported(int B, int N, int C0, int H, int nl, int Co)
{
    return sn_fc_chain_forward_pool_supported(B, N, C0, H, nl) && Co >= 32 && Co <= H && Co % 32 == 0;
}

extern "C" int sn_fc_chain_forward_pool_out(int B, int N, int nconv, long long *acc, const float *pool_val, const int *pool_idx,
                                            const float *gamma5, const float *beta5, float *running_mean5, float *running_var5,
                                            long long *num_batches_tracked5, float eps5, float momentum5, float *coef5, float *pooled,
                                            int *argsel, float *zsel, int H, int nl, const float *const *W, const float *const *bias,
                                            const float *const *gamma, const float *const *beta, float *const *running_mean,
                                            float *const *running_var, long long *const *num_batches_tracked, const float *eps,
                                            const float *momentum, float *const *z, float *const *coef, float *xbuf, unsigned *sync,
                                            int Co, const float *Wo, const float *bo, const float *gamma_o, const float *beta_o,
                                            float *running_mean_o, float *running_var_o, long long *num_batches_tracked_o, float eps_o,
                                            float momentum_o, float *zo, float *coef_o, float *y, sn_stream_t stream)
{
    constexpr int C0 = 128;
    SN_REQUIRE(sn_fc_chain_forward_pool_out_supported(B, N, C0, H, nl, Co) && nconv >= 2, "shape not supported (sn_fc_chain_forward_pool_out_supported)");
    SN_REQUIRE(acc && pool_val && pool_idx && gamma5 && beta5 && running_mean5 && running_var5 && coef5 && pooled && argsel && zsel,
               "null pointer");
    SN_REQUIRE(W && bias && gamma && beta && eps && momentum && z && coef && xbuf && sync, "null pointer");
    SN_REQUIRE(Wo && bo && gamma_o && beta_o && zo && coef_o && y, "null output-layer pointer");
    FcChainArgs g{};
    g.a0 = pooled, g.R = B, g.C0 = C0, g.H = H, g.nl = nl, g.xbuf = xbuf, g.sync = sync;
    g.P.acc = acc + (size_t)(nconv - 1) * kFxLayer, g.P.zero_ptr = acc + (size_t)(nconv - 2) * kFxLayer, g.P.zero_n = kFxLayer;
    g.P.keys = reinterpret_cast<const unsigned long long *>(pool_val);
    (void)pool_idx;
    g.P.bn = BnFwd{gamma5, beta5, running_mean5, running_var5, num_batches_tracked5, coef5, eps5, momentum5, (long long)B * N};
    g.P.pooled = pooled, g.P.argsel = argsel, g.P.zsel = zsel;
    for (int l = 0; l < nl; ++l) {
        SN_REQUIRE(W[l] && bias[l] && gamma[l] && beta[l] && z[l] && coef[l], "null layer pointer");
        g.L[l] = FcChainLayer{W[l], bias[l], gamma[l], beta[l], running_mean ? running_mean[l] : nullptr,
                              running_var ? running_var[l] : nullptr, num_batches_tracked ? num_batches_tracked[l] : nullptr,
                              z[l], coef[l], eps[l], momentum[l]};
    }
    g.O = FcChainOut{Wo, bo, gamma_o, beta_o, running_mean_o, running_var_o, num_batches_tracked_o, zo, coef_o, y, eps_o, momentum_o, Co};
    const size_t lds = fc_chain_fwd_lds(C0, H, nl);
    static SnLdsAttrGrow attr;
    if (sn_lds_attr_grow(attr, (const void *)fc_chain_fwd_kernel<128, 3, true, true>, lds, "sn_fc_chain_forward_pool_out")) return SN_ERR_UNSUPPORTED;
    g.rinv_rows = 1.0 / (double)B, g.unbias = B > 1 ? (double)B / (double)(B - 1) : 1.0;
    hipLaunchKernelGGL((fc_chain_fwd_kernel<128, 3, true, true>), dim3(8 * (H / 32)), dim3(256), lds, (hipStream_t)stream, g);
    SN_LAUNCH_CHECK();
    return 0;
}

// (sn_fc_chain_backward_obn hands the output BatchNorm's operands to the launch below: thread-local, consumed by the next call)
struct FcBwdOutBn {
    const float *z, *coef;
    float *dgamma, *dbeta;
    int fixed;
};
static thread_local FcBwdOutBn g_obn{};

// 1: sn_fc_chain_backward runs this FC head's backward (ns GEMM layers, top first: Co[s] x Ci[s]) as one launch
extern "C" int sn_fc_chain_backward_supported(int R, int ns, const int *Co, const int *Ci)
{
    if (R < 1 || R > 32 || ns < 2 || ns > kFcBwdMaxStages || !Co || !Ci) return 0;
    for (int s = 0; s < ns; ++s) {
        if (Co[s] < 64 || Co[s] > 256 || Co[s] % 64 || Ci[s] < 32 || Ci[s] > 256 || Ci[s] % 32) return 0;
        if (s > 0 && Co[s] != Ci[s - 1]) return 0;
        if (s + 1 < ns && Ci[s] != 256) return 0;  // the hand-off slabs and the gather are sized for 256 columns
    }
    return 1;
}

extern "C" int sn_fc_chain_backward(int R, int ns, const int *Co, const int *Ci, const float *gy, const float *const *W,
                                    const float *const *zprev, const float *const *coefprev, const long long *bn_rows,
                                    float *const *dgamma, float *const *dbeta, float *const *dbias, float *const *dW, float *db_top,
                                    const float *const *aprev, const int *araw, float *gout, float *kout, float *xbuf,
                                    unsigned *sync, sn_stream_t stream)
{
    SN_REQUIRE(sn_fc_chain_backward_supported(R, ns, Co, Ci), "shape not supported (sn_fc_chain_backward_supported)");
    SN_REQUIRE(gy && W && zprev && coefprev && bn_rows && dgamma && dbeta && dbias && dW && aprev && araw && xbuf && sync, "null pointer");
    FcBwdArgs g{};
    g.gy = gy, g.R = R, g.ns = ns, g.xbuf = xbuf, g.sync = sync;
    for (int s = 0; s < ns; ++s) {
        SN_REQUIRE(W[s] && zprev[s] && coefprev[s] && dgamma[s] && dbeta[s] && dW[s] && aprev[s], "null stage pointer");
        FcBwdStage &S = g.S[s];
        S.W = W[s], S.Co = Co[s], S.Ci = Ci[s], S.zprev = zprev[s], S.coefprev = coefprev[s], S.bn_rows = bn_rows[s];
        S.rinv = bn_rows[s] > 0 ? 1.0 / (double)bn_rows[s] : 0.0;
        S.dgamma = dgamma[s], S.dbeta = dbeta[s], S.dbias = dbias[s], S.dW = dW[s], S.db = s == 0 ? db_top : nullptr;
        S.aprev = aprev[s], S.araw = araw[s];
        S.gout = s == ns - 1 ? gout : nullptr, S.kout = s == ns - 1 ? kout : nullptr;
    }
    if (g_obn.z) g.oz = g_obn.z, g.ocoef = g_obn.coef, g.odgamma = g_obn.dgamma, g.odbeta = g_obn.dbeta, g.ofixed = g_obn.fixed;
    g_obn = FcBwdOutBn{};
    const size_t lds = ((size_t)3 * 32 * 260 + kRsFloats + 32 * 36) * sizeof(float);
    static SnLdsAttr attr;
    if (sn_lds_attr(attr, (const void *)fc_chain_bwd_kernel, lds, "sn_fc_chain_backward")) return SN_ERR_UNSUPPORTED;
    // 128 blocks: b % 8 == 0 -> XCD 0; b / 8 = 0..7 the chain, 8..15 the weight-gradient workgroups
    hipLaunchKernelGGL(fc_chain_bwd_kernel, dim3(128), dim3(256), lds, (hipStream_t)stream, g);
    SN_LAUNCH_CHECK();
    return 0;
}

// sn_fc_chain_backward for a head whose output went through a BatchNorm WITHOUT activation (classification sampler: sn_layer_forward_bn_out
// / sn_fc_chain_forward_pool_out): gy is the gradient behind that BatchNorm, zo (R, Co[0]) its input and coef_o (4, Co[0]) the forward's
// coefficients; the launch opens with its backward (as sn_bn_output_backward) and leaves dgamma_o / dbeta_o (Co[0]).  fixed != 0: the
// forward ran on running statistics.
extern "C" int sn_fc_chain_backward_obn(int R, int ns, const int *Co, const int *Ci, const float *gy, const float *zo, const float *coef_o,
                                        int fixed, float *dgamma_o, float *dbeta_o, const float *const *W, const float *const *zprev,
                                        const float *const *coefprev, const long long *bn_rows, float *const *dgamma, float *const *dbeta,
                                        float *const *dbias, float *const *dW, float *db_top, const float *const *aprev, const int *araw,
                                        float *gout, float *kout, float *xbuf, unsigned *sync, sn_stream_t stream)
{
    SN_REQUIRE(zo && coef_o && dgamma_o && dbeta_o, "null output-BatchNorm pointer");
    g_obn = FcBwdOutBn{zo, coef_o, dgamma_o, dbeta_o, fixed ? 1 : 0};
    const int rc = sn_fc_chain_backward(R, ns, Co, Ci, gy, W, zprev, coefprev, bn_rows, dgamma, dbeta, dbias, dW, db_top, aprev, araw, gout,
                                        kout, xbuf, sync, stream);
    g_obn = FcBwdOutBn{};
    return rc;
}
