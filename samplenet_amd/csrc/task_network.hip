// task_network.hip -- the task network behind the sampler (registration/models/pcrnet.py:23-77: PointNetFeatures, a BatchNorm-free
// 3 -> 64 -> 64 -> 64 -> 128 -> 1024 ReLU extractor with a max over the points, and the 2048 -> ... -> 7 trunk on at most 32 rows)
// on the split-bf16 GEMM core of mlp_device.h: the wide last layer with the max-pool in its epilogue, the narrow front as one launch
// per direction, the skinny trunk layers (forward, data gradient, weight gradient) and the sparse data gradient through the max.
#include "mlp_device.h"

using namespace sn;

// ---- the same layer as a WIDE product (Co >> 64: PCRNet's 128 -> 1024 on 32 x 1024 points is 8.6 GFLOP, 16 column blocks per row
// tile).  linear_fwd_kernel runs it as 512 x 16 independent 64 x 64 tiles: every tile re-reads, re-activates and re-splits its A
// rows and re-splits its W block (86 us = 100 fp32-equivalent TFLOP/s).  Here a workgroup of four waves owns 128 rows for ALL of
// its columns: each wave activates and splits its 32 rows ONCE into the three bf16 planes' A fragments, which then stay in
// registers (3 x K/16 x 16 bytes per lane); the weights arrive pre-split (split_planes_kernel, once per call) and only their
// 64-column blocks move through LDS, double-buffered -- the next block's global loads are in flight under the current
// block's 48 MFMAs per wave, one barrier per block.  Same six products per 16 k in the same order as gemm_tile_bx3: the
// pre-activations, hence the pooled features, are bit-identical to linear_fwd_kernel's.  Epilogue per block: bias, (max, first
// row) over the wave's 32 rows as a 64-bit key (a wave's rows lie in one cloud: npts % 32 == 0), the waves of one cloud combined
// through LDS, ONE plain 8-byte store per column and min(128, npts) rows; the decode kernel takes the maximum over a cloud's
// npts / 128 keys.  No atomics (one atomicMax per column and 32 rows = 1 M of them per call paced the kernel at 70 us whatever the
// MFMAs did), no key clear.  Small R: the columns are split over gridDim.y so that the grid still covers the chip.
constexpr int kWideRows = 128, kWideBN = 64;
struct WideArgs {
    const float *ain, *scale, *shift;  // (R, K) pre-activations of the layer below and its operand coefficients (NULL: identity)
    const __bf16 *planes;               // [3][Co][K]
    const float *bias;
    float *z;                           // (R, Co) or NULL
    unsigned long long *partial;        // [R / group_rows][Co]: (max, first row) keys of group_rows = min(128, npts) rows
    int R, Co, npts, cols_per_wg, group_rows;
};
__global__ void __launch_bounds__(256) split_planes_kernel(int n, const float *__restrict__ W, __bf16 *__restrict__ planes)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    __bf16 h1, h2, h3;
    split3(W[i], h1, h2, h3);
    planes[i] = h1, planes[(size_t)n + i] = h2, planes[2 * (size_t)n + i] = h3;
}
// STORE_Z: the pre-activations are written too (a backward through trainable weights will read them); ARG: the keys carry the row of
// the maximum (only a backward needs it).  Software pipeline: the epilogue of block k (bias, maximum over the wave's rows, key) is
// spread over the eight k-steps of block k + 1's MFMAs -- two waves share a SIMD and all eight meet at a barrier every block, so
// an epilogue phase of its own is a phase in which no matrix instruction issues anywhere on the CU (measured: 63 us of which 31
// were MFMA time); as fillers between MFMAs the same instructions are nearly free (MI355X_MICROARCH.md: <= 5 per gap).
#ifndef SN_WIDE_X
#define SN_WIDE_X 0  // (timing experiments: 1 no sched_barriers, 2 static priority for waves 4-7, 4 no epilogue fillers, 5 no weight-block traffic)
#endif
template <int K, bool STORE_Z, bool ARG>
__global__ void __launch_bounds__(512) linear_fwd_wide_pool_kernel(WideArgs g)
{
    constexpr int KS = K / 16, PITCH = K + 8;            // 16 consecutive rows' 16-byte fragments tile all 64 banks (K % 32 == 0)
    constexpr int NB = 3 * kWideBN * (K / 8) / 512;      // 16-byte items of a weight block per thread
    constexpr int BUF = 3 * kWideBN * PITCH;             // bf16 elements per buffer
    constexpr int EPK = 16 / KS;                         // epilogue elements per k-step
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ unsigned long long s_keys[2][4][kWideBN];
    __bf16 *Bs = reinterpret_cast<__bf16 *>(lds);
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // eight waves: row tile rw = wave & 3 (32 rows each), column half cw = wave >> 2 of every 64-column block
    const int rw = wave & 3, cw = wave >> 2;
    const int row0 = blockIdx.x * kWideRows + rw * 32;
    const int cbeg = blockIdx.y * g.cols_per_wg, nblk = g.cols_per_wg / kWideBN;
    const int Co = g.Co;
    // per-thread item offsets of a weight block, fixed for the whole kernel: global (elements from the block's first column's row)
    // and LDS (elements from the buffer)
    int goff[NB], loff[NB];
#pragma unroll
    for (int q = 0; q < NB; ++q) {
        const int f = tid + q * 512, p = f / (kWideBN * (K / 8)), r = f % (kWideBN * (K / 8)), x = r / (K / 8), k8 = (r % (K / 8)) * 8;
        goff[q] = (p * Co + x) * K + k8;
        loff[q] = (p * kWideBN + x) * PITCH + k8;
    }
    bf16x8 rb[NB];
    const auto fetch_b = [&](int col0) {
        const __bf16 *pb = g.planes + (size_t)col0 * K;  // (uniform)
#pragma unroll
        for (int q = 0; q < NB; ++q) rb[q] = *reinterpret_cast<const bf16x8 *>(pb + goff[q]);
    };
    const auto stage_b = [&](__bf16 *buf) {
#pragma unroll
        for (int q = 0; q < NB; ++q) *reinterpret_cast<bf16x8 *>(buf + loff[q]) = rb[q];
    };
    SN_TL(0);
    // Column blocks are visited in an order rotated by the workgroup's row block: all workgroups run in step, and 256 of them
    // asking the L2 for the SAME 48 KB at the same moment serialise on the few channels those lines live in
    const int rot = (blockIdx.x / 8) % nblk;  // (workgroups b, b + 8, ... share an XCD and its L2)
    const auto blk_col = [&](int blk) { return cbeg + ((blk + rot) % nblk) * kWideBN; };
    fetch_b(blk_col(0));
    // this wave's A fragments: lane -> row l31, 8 consecutive k at 16 kk + 8 h; activated and split once
    bf16x8 a[3][KS];
    {
        const float *ar = g.ain + (size_t)(row0 + l31) * K + 8 * h;
        float4 v[KS][2];
#pragma unroll
        for (int kk = 0; kk < KS; ++kk)
            v[kk][0] = *reinterpret_cast<const float4 *>(ar + kk * 16), v[kk][1] = *reinterpret_cast<const float4 *>(ar + kk * 16 + 4);
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
            float e[8] = {v[kk][0].x, v[kk][0].y, v[kk][0].z, v[kk][0].w, v[kk][1].x, v[kk][1].y, v[kk][1].z, v[kk][1].w};
            if (g.scale) {
                const float4 s0 = *reinterpret_cast<const float4 *>(g.scale + kk * 16 + 8 * h), s1 = *reinterpret_cast<const float4 *>(g.scale + kk * 16 + 8 * h + 4);
                const float4 t0 = *reinterpret_cast<const float4 *>(g.shift + kk * 16 + 8 * h), t1 = *reinterpret_cast<const float4 *>(g.shift + kk * 16 + 8 * h + 4);
                const float sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w}, sh[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
#pragma unroll
                for (int t = 0; t < 8; ++t) e[t] = relu_np(fmaf(e[t], sc[t], sh[t]));
            }
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                __bf16 h1, h2, h3;
                split3(e[t], h1, h2, h3);
                a[0][kk][t] = h1, a[1][kk][t] = h2, a[2][kk][t] = h3;
            }
        }
    }
    stage_b(Bs);
    __syncthreads();
    SN_TL(1);
    if (SN_WIDE_X == 2 && wave >= 4) __builtin_amdgcn_s_setprio(1);
    const int rin0 = row0 % g.npts;                    // this wave's first row inside its cloud
    const int wpg = g.group_rows / 32, ngrp = 4 / wpg;  // row waves per key group, key groups per workgroup
    const int boff = (cw * 32 + l31) * PITCH + 8 * h;   // this lane's B fragment inside a plane of a buffer (k-step 0)
    float *zrow = STORE_Z ? g.z + (size_t)(row0 + 4 * h) * Co + cw * 32 + l31 : nullptr;  // + frag rows, + col0
    f32x16 accp;      // the previous block's accumulators, epilogue pending
    float biasp = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) accp[e] = 0.f;
    // iteration blk: MFMAs of block blk (blk < nblk) + epilogue of block blk - 1 (blk > 0); the keys of block blk - 1 are
    // published by the barrier that ends iteration blk and stored right behind it
    for (int blk = 0; blk <= nblk; ++blk) {
        const bool mm = blk < nblk, ep = blk > 0;
        if (blk == 4) SN_TL(2);
        const __bf16 *cur = Bs + (blk & 1) * BUF + boff;
        const int col0 = blk_col(blk), colp = blk_col(blk - 1 + nblk);  // this block's first column, the previous block's
        if (blk + 1 < nblk && SN_WIDE_X != 5) fetch_b(blk_col(blk + 1));
        const float biasv = (mm && g.bias) ? g.bias[col0 + cw * 32 + l31] : 0.f;
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
        bf16x8 b[2][3];
        const auto load_b = [&](int kk, bf16x8 (&bb)[3]) {
#pragma unroll
            for (int p = 0; p < 3; ++p) bb[p] = *reinterpret_cast<const bf16x8 *>(cur + p * kWideBN * PITCH + kk * 16);
        };
        float m = -INFINITY;
        int im = 0;
        if (mm) load_b(0, b[0]);
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
            if (mm) {
                if (kk + 1 < KS) load_b(kk + 1, b[(kk + 1) & 1]);
#define SN_WIDE_TERM(PA, PB) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA][kk], b[kk & 1][PB], acc, 0, 0, 0)
                // the six MFMAs of a k-step run on ONE accumulator: kept back to back (a filler between two of them costs ~43 cycles,
                // MI355X_MICROARCH.md); the fragment reads and the epilogue pieces go between the groups
                if (SN_WIDE_X != 1) __builtin_amdgcn_sched_barrier(0);
                SN_WIDE_TERM(0, 2);
                SN_WIDE_TERM(2, 0);
                SN_WIDE_TERM(1, 1);
                SN_WIDE_TERM(0, 1);
                SN_WIDE_TERM(1, 0);
                SN_WIDE_TERM(0, 0);
                if (SN_WIDE_X != 1) __builtin_amdgcn_sched_barrier(0);
#undef SN_WIDE_TERM
            }
            // the next block's weights (requested at the top of the iteration) go to the other buffer under the last MFMA groups; that
            // buffer's readers passed the barrier that ended iteration blk - 1
            if (kk == KS - 2 && blk + 1 < nblk && SN_WIDE_X != 5) stage_b(Bs + ((blk + 1) & 1) * BUF);
            if (ep && SN_WIDE_X != 4) {
#pragma unroll
                for (int e = kk * EPK; e < (kk + 1) * EPK; ++e) {  // rows ascend with e inside a lane: strict compare = first occurrence
                    const float v = accp[e] + biasp;
                    if (STORE_Z) zrow[(size_t)((e & 3) + 8 * (e >> 2)) * Co + colp] = v;
                    if (ARG) {
                        if (v > m) m = v, im = (e & 3) + 8 * (e >> 2);
                    } else {
                        m = fmaxf(m, v);
                    }
                }
            }
        }
        if (ep) {
            if (ARG) im += 4 * h;
            const float om = __shfl_xor(m, 32);
            const int oim = __shfl_xor(im, 32);
            if (om > m || (ARG && om == m && oim < im)) m = om, im = oim;
            if (lane < 32) s_keys[blk & 1][rw][cw * 32 + l31] = pool_key(m, ARG ? rin0 + im : 0);
        }
        if (blk == 4) SN_TL(3);
        accp = acc, biasp = biasv;
        // LDS-only barrier (__syncthreads() would also wait for the block's global stores)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (blk == 4) SN_TL(5);
        if (ep && tid < ngrp * kWideBN) {  // (s_keys[blk & 1] is rewritten two iterations on, behind another barrier)
            const int grp = tid / kWideBN, c = tid % kWideBN;
            unsigned long long k = s_keys[blk & 1][grp * wpg][c];
            for (int w = 1; w < wpg; ++w) k = max(k, s_keys[blk & 1][grp * wpg + w][c]);
            g.partial[((size_t)blockIdx.x * ngrp + grp) * Co + colp + c] = k;
        }
    }
    SN_TL(6);
}
// pooled = relu(max over the cloud's P partial keys), the row and the pre-activation value (what the pooling backward needs)
__global__ void __launch_bounds__(256) maxpool_partials_decode_kernel(int n, int Co, int P, const unsigned long long *__restrict__ partial,
                                                                      float *__restrict__ pooled, int *__restrict__ argsel,
                                                                      float *__restrict__ zsel)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int b = i / Co, c = i % Co;
    unsigned long long k = partial[(size_t)b * P * Co + c];
    for (int p = 1; p < P; ++p) k = max(k, partial[((size_t)b * P + p) * Co + c]);
    float v;
    int row;
    pool_key_decode(k, v, row);
    pooled[i] = relu_np(v);
    if (argsel) argsel[i] = row;
    if (zsel) zsel[i] = v;
}

extern "C" int sn_linear_forward_maxpool_wide_supported(int R, int Ci, int Co, int npts)
{
    // (below 128 row blocks the columns would have to be split four ways and more to cover the chip: the per-workgroup prologue
    // -- activate and split 128 rows, ~6 us -- then outweighs what the 64 x 64 tile kernel re-does per tile)
    return R >= 128 * kWideRows && R % kWideRows == 0 && npts >= 32 && npts % 32 == 0 && R % npts == 0 && (Ci == 64 || Ci == 128) &&
           Co >= 8 * kWideBN && Co % kWideBN == 0;
}
static int wide_group_rows(int npts) { return npts % 128 == 0 ? 128 : npts % 64 == 0 ? 64 : 32; }
extern "C" long long sn_linear_forward_maxpool_wide_scratch_bytes(int R, int Ci, int Co, int npts)
{
    (void)Ci;
    return (long long)(R / wide_group_rows(npts)) * Co * (long long)sizeof(unsigned long long);
}
// wplanes: 3 * Co * Ci bf16 for the split weights; planes_ready != 0: it already holds the split of THIS W (a second cloud
// through the same frozen layer).  scratch: _scratch_bytes (the per-group keys).
extern "C" int sn_linear_forward_maxpool_wide(int R, int Ci, int Co, int npts, const float *ain, const float *coef_prev, const float *W,
                                              const float *bias, float *z, void *scratch, float *pooled, int *argsel, float *zsel,
                                              void *wplanes, int planes_ready, sn_stream_t stream)
{
    SN_REQUIRE(ain && W && scratch && pooled && wplanes, "null pointer");
    if (!sn_linear_forward_maxpool_wide_supported(R, Ci, Co, npts))
        return sn_set_error(SN_ERR_UNSUPPORTED, "sn_linear_forward_maxpool_wide: needs R %% 128 == 0, npts %% 32 == 0, Ci 64 / 128, Co %% 64 == 0");
    hipStream_t st = (hipStream_t)stream;
    const int B = R / npts;
    __bf16 *planes = (__bf16 *)wplanes;
    if (!planes_ready) hipLaunchKernelGGL(split_planes_kernel, dim3((Co * Ci + 255) / 256), dim3(256), 0, st, Co * Ci, W, planes);
    WideArgs g{};
    g.ain = ain, g.scale = coef_prev, g.shift = coef_prev ? coef_prev + Ci : nullptr;
    g.planes = planes, g.bias = bias, g.z = z, g.partial = (unsigned long long *)scratch, g.R = R, g.Co = Co, g.npts = npts;
    g.group_rows = wide_group_rows(npts);
    // columns per workgroup: all of them when the row blocks alone cover the chip, else split (a power-of-two number of 64-column blocks)
    const int rb = R / kWideRows;
    int cs = 1;
    while (rb * cs < 256 && Co / (cs * 2) >= kWideBN && (Co / kWideBN) % (cs * 2) == 0) cs *= 2;
    g.cols_per_wg = Co / cs;
    const size_t lds = (size_t)2 * 3 * kWideBN * (Ci + 8) * sizeof(__bf16);
    const bool sz = z != nullptr, arg = argsel != nullptr || sz;
#define SN_WIDE_LAUNCH(KK, SZ, AR)                                                                                                  \
    do {                                                                                                                            \
        static SnLdsAttr attr;                                                                                                      \
        if (sn_lds_attr(attr, (const void *)linear_fwd_wide_pool_kernel<KK, SZ, AR>, (size_t)2 * 3 * kWideBN * (KK + 8) * 2,        \
                        "sn_linear_forward_maxpool_wide"))                                                                          \
            return SN_ERR_UNSUPPORTED;                                                                                              \
        hipLaunchKernelGGL((linear_fwd_wide_pool_kernel<KK, SZ, AR>), dim3(rb, cs), dim3(512), lds, st, g);                         \
    } while (0)
    if (Ci == 128) {
        if (sz) SN_WIDE_LAUNCH(128, true, true); else if (arg) SN_WIDE_LAUNCH(128, false, true); else SN_WIDE_LAUNCH(128, false, false);
    } else {
        if (sz) SN_WIDE_LAUNCH(64, true, true); else if (arg) SN_WIDE_LAUNCH(64, false, true); else SN_WIDE_LAUNCH(64, false, false);
    }
#undef SN_WIDE_LAUNCH
    hipLaunchKernelGGL(maxpool_partials_decode_kernel, dim3((B * Co + 255) / 256), dim3(256), 0, st, B * Co, Co, npts / g.group_rows,
                       (const unsigned long long *)scratch, pooled, argsel, zsel);
    SN_LAUNCH_CHECK();
    return 0;
}

// ---- the narrow front of a BatchNorm-free extractor in ONE launch: 3 -> 64 -> 64 -> 64 -> 128 with ReLU between (PCRNet's
// PointNetFeatures conv1..conv4, registration/models/pcrnet.py:23-38).  Without a BatchNorm a row's layers depend on nothing but
// the row: a wave takes 32 rows from the three coordinates to the 128 pre-activations of conv4 -- conv1 on the VALU straight into
// the A fragments of conv2 (conv_in3_fwd_kernel's expression), conv2..conv4 as split-bf16 MFMAs against weight planes staged
// once per workgroup in LDS (110 KB), each layer's 32 x 64 output tile turned from the accumulator layout (lane = column) into the
// next layer's fragment layout (lane = row) through a wave-private LDS tile, activated and split on the way.  Same products in
// the same order as conv_in3_fwd_kernel / linear_fwd_kernel's gemm_tile_bx3: every layer's pre-activations are bit-identical to
// the layer-by-layer launches.  z1..z3 are written only when a backward will read them (NULL otherwise): the frozen template
// branch reads 12 bytes per point and writes conv4's 512.  (4 launches of 5-10 us each before.)
constexpr int kNarrowRows = 128;
struct NarrowArgs {
    const float *x, *W1, *b1, *b2, *b3, *b4;
    const __bf16 *P2, *P3, *P4;  // [3][64][64], [3][64][64], [3][128][64]
    float *z1, *z2, *z3, *z4;
    int R;
    unsigned long long *zero_keys;  // rider: a scratch the NEXT launch wants cleared (the pooled layer's per-cloud keys), zero_n words
    int zero_n;
};
struct SplitJob3 {
    const float *w[3];
    __bf16 *dst[3];
    int n[3];
};
__global__ void __launch_bounds__(256) split_planes3_kernel(SplitJob3 job)
{
    const int l = blockIdx.y, n = job.n[l];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    __bf16 h1, h2, h3;
    split3(job.w[l][i], h1, h2, h3);
    job.dst[l][i] = h1, job.dst[l][(size_t)n + i] = h2, job.dst[l][2 * (size_t)n + i] = h3;
}
template <bool STORE>
__global__ void __launch_bounds__(256) pointnet_narrow_fwd_kernel(NarrowArgs g)
{
    constexpr int PW = 72;   // plane row pitch (bf16): K = 64 + 8 -- a b128 lane group's 16 rows tile all 64 banks
    constexpr int PT = 68;   // transpose tile pitch (floats)
    constexpr int N2 = 3 * 64 * PW, N4 = 3 * 128 * PW;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __bf16 *S2 = reinterpret_cast<__bf16 *>(lds), *S3 = S2 + N2, *S4 = S3 + N2;
    float *W1s = reinterpret_cast<float *>(S4 + N4);  // [64][4] = (w0, w1, w2, b)
    float *Tall = W1s + 64 * 4;                       // [4 waves][32][PT]
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float *T = Tall + wave * (32 * PT);
    SN_TL(0);
    if (g.zero_keys)
        for (int i = blockIdx.x * 256 + tid; i < g.zero_n; i += gridDim.x * 256) g.zero_keys[i] = 0ull;
    // stage the planes (straight 16-byte copies: global [3][Co][64] -> LDS [3][Co][PW]) and conv1's weights
    {
        // (all 24 loads of a thread in flight before the first LDS store: one memory round trip, not one per item)
        bf16x8 r2[6], r3[6], r4[12];
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            const int f = tid + q * 256;
            r2[q] = *reinterpret_cast<const bf16x8 *>(g.P2 + (size_t)f * 8), r3[q] = *reinterpret_cast<const bf16x8 *>(g.P3 + (size_t)f * 8);
        }
#pragma unroll
        for (int q = 0; q < 12; ++q) r4[q] = *reinterpret_cast<const bf16x8 *>(g.P4 + (size_t)(tid + q * 256) * 8);
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            const int f = tid + q * 256, row = f >> 3, k8 = (f & 7) * 8;  // row = plane * Co + column; global rows are 64 wide
            *reinterpret_cast<bf16x8 *>(S2 + row * PW + k8) = r2[q];
            *reinterpret_cast<bf16x8 *>(S3 + row * PW + k8) = r3[q];
        }
#pragma unroll
        for (int q = 0; q < 12; ++q) {
            const int f = tid + q * 256, row = f >> 3, k8 = (f & 7) * 8;
            *reinterpret_cast<bf16x8 *>(S4 + row * PW + k8) = r4[q];
        }
        if (tid < 64) {
            W1s[tid * 4 + 0] = g.W1[tid * 3 + 0], W1s[tid * 4 + 1] = g.W1[tid * 3 + 1], W1s[tid * 4 + 2] = g.W1[tid * 3 + 2];
            W1s[tid * 4 + 3] = g.b1 ? g.b1[tid] : 0.f;
        }
    }
    const int row0 = blockIdx.x * kNarrowRows + wave * 32;
    const int rrow = min(row0 + l31, g.R - 1);
    const float x0 = g.x[(size_t)rrow * 3], x1 = g.x[(size_t)rrow * 3 + 1], x2 = g.x[(size_t)rrow * 3 + 2];
    __syncthreads();
    SN_TL(1);
    const bool rok = row0 + l31 < g.R;
    // conv1 straight into conv2's A fragments: lane -> row l31, channels 16 kk + 8 h + t
    bf16x8 a[3][4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        float e[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
#pragma clang fp contract(off)
            const float4 w = *reinterpret_cast<const float4 *>(W1s + (kk * 16 + 8 * h + t) * 4);
            e[t] = fmaf(w.z, x2, fmaf(w.y, x1, w.x * x0)) + w.w;  // (conv_in3_fwd_kernel's expression, bit for bit)
        }
        if (STORE && rok) {
            float *zp = g.z1 + (size_t)(row0 + l31) * 64 + kk * 16 + 8 * h;
            *reinterpret_cast<float4 *>(zp) = make_float4(e[0], e[1], e[2], e[3]);
            *reinterpret_cast<float4 *>(zp + 4) = make_float4(e[4], e[5], e[6], e[7]);
        }
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            __bf16 h1, h2, h3;
            split3(relu_np(e[t]), h1, h2, h3);
            a[0][kk][t] = h1, a[1][kk][t] = h2, a[2][kk][t] = h3;
        }
    }
    // one 64-wide layer: acc (two 32-column tiles) from the fragments a[][] and the planes S; bias; optional store; through T into
    // the next layer's fragments
    const auto layer64 = [&](const __bf16 *S, const float *bias, float *zout) {
        f32x16 acc[2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
        // (all of the layer's B fragments requested before its first MFMA: 24 LDS reads in flight instead of six at a time in
        // front of every k-step -- one wave per SIMD, nothing else hides an LDS read)
        bf16x8 b[4][3][2];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int j = 0; j < 2; ++j) b[kk][p][j] = *reinterpret_cast<const bf16x8 *>(S + (p * 64 + j * 32 + l31) * PW + kk * 16 + 8 * h);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
#define SN_NR_TERM(PA, PB) _Pragma("unroll") for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA][kk], b[kk][PB][j], acc[j], 0, 0, 0)
            SN_NR_TERM(0, 2);
            SN_NR_TERM(2, 0);
            SN_NR_TERM(1, 1);
            SN_NR_TERM(0, 1);
            SN_NR_TERM(1, 0);
            SN_NR_TERM(0, 0);
#undef SN_NR_TERM
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = j * 32 + l31;
            const float bv = bias ? bias[n] : 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int r = frag_row(e, lane);
                T[r * PT + n] = acc[j][e] + bv;
            }
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const float4 v0 = *reinterpret_cast<const float4 *>(T + l31 * PT + kk * 16 + 8 * h), v1 = *reinterpret_cast<const float4 *>(T + l31 * PT + kk * 16 + 8 * h + 4);
            if (STORE && rok) {  // (16-byte stores from the row layout: dword stores from the accumulator layout are issue-bound)
                float *zp = zout + (size_t)(row0 + l31) * 64 + kk * 16 + 8 * h;
                *reinterpret_cast<float4 *>(zp) = v0, *reinterpret_cast<float4 *>(zp + 4) = v1;
            }
            const float e[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                __bf16 h1, h2, h3;
                split3(relu_np(e[t]), h1, h2, h3);
                a[0][kk][t] = h1, a[1][kk][t] = h2, a[2][kk][t] = h3;
            }
        }
        __builtin_amdgcn_wave_barrier();
    };
    SN_TL(2);
    layer64(S2, g.b2, g.z2);
    SN_TL(3);
    layer64(S3, g.b3, g.z3);
    SN_TL(4);
    // conv4: 128 columns, straight to memory in the accumulator layout (a lane's column, two rows per instruction: 128-byte runs)
    {
        f32x16 acc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
        bf16x8 b[2][3][4];  // one k-step ahead
        const auto load_b4 = [&](int kk, bf16x8 (&bb)[3][4]) {
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int j = 0; j < 4; ++j) bb[p][j] = *reinterpret_cast<const bf16x8 *>(S4 + (p * 128 + j * 32 + l31) * PW + kk * 16 + 8 * h);
        };
        load_b4(0, b[0]);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            if (kk + 1 < 4) load_b4(kk + 1, b[(kk + 1) & 1]);
#define SN_NR_TERM(PA, PB) _Pragma("unroll") for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA][kk], b[kk & 1][PB][j], acc[j], 0, 0, 0)
            SN_NR_TERM(0, 2);
            SN_NR_TERM(2, 0);
            SN_NR_TERM(1, 1);
            SN_NR_TERM(0, 1);
            SN_NR_TERM(1, 0);
            SN_NR_TERM(0, 0);
#undef SN_NR_TERM
        }
        SN_TL(5);
        // through the wave's tile in two 64-column halves, out as 16-byte stores (4 rows x 256 bytes per instruction): 64 dword
        // stores per lane from the accumulator layout took 8 of the kernel's 17 us (store-issue-bound)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const int j = hf * 2 + jj, n = j * 32 + l31;
                const float bv = g.b4 ? g.b4[n] : 0.f;
#pragma unroll
                for (int e = 0; e < 16; ++e) T[frag_row(e, lane) * PT + jj * 32 + l31] = acc[j][e] + bv;
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int r = (lane >> 4) + 4 * i, c4 = (lane & 15) * 4;
                const float4 v = *reinterpret_cast<const float4 *>(T + r * PT + c4);
                if (row0 + r < g.R) *reinterpret_cast<float4 *>(g.z4 + (size_t)(row0 + r) * 128 + hf * 64 + c4) = v;
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    SN_TL(6);
}

extern "C" int sn_pointnet_narrow_forward_supported(int R, int c1, int c2, int c3, int c4)
{
    // (whole 64-row tiles: the layer-by-layer route runs ragged tiles on the fp32 MFMA, and the two routes are to stay bit-identical)
    return R >= 64 && R % 64 == 0 && c1 == 64 && c2 == 64 && c3 == 64 && c4 == 128;
}
// The narrow front 3 -> 64 -> 64 -> 64 -> 128 (weights (64,3), (64,64), (64,64), (128,64), biases optional).  wplanes: 3 * (64*64 + 64*64 +
// 128*64) bf16 of scratch for the split weights; planes_ready != 0: it already holds the split of THESE weights.  z1..z3 (R,64):
// all three or none (NULL: not written); z4 (R,128).
extern "C" int sn_pointnet_narrow_forward(int R, const float *x, const float *W1, const float *b1, const float *W2, const float *b2,
                                          const float *W3, const float *b3, const float *W4, const float *b4, void *wplanes,
                                          int planes_ready, float *z1, float *z2, float *z3, float *z4, unsigned long long *zero_keys,
                                          int zero_n, sn_stream_t stream)
{
    SN_REQUIRE(R >= 1 && x && W1 && W2 && W3 && W4 && wplanes && z4, "bad argument");
    SN_REQUIRE((z1 && z2 && z3) || (!z1 && !z2 && !z3), "z1..z3: all three or none");
    hipStream_t st = (hipStream_t)stream;
    __bf16 *P2 = (__bf16 *)wplanes, *P3 = P2 + 3 * 64 * 64, *P4 = P3 + 3 * 64 * 64;
    if (!planes_ready) {
        SplitJob3 job{{W2, W3, W4}, {P2, P3, P4}, {64 * 64, 64 * 64, 128 * 64}};
        hipLaunchKernelGGL(split_planes3_kernel, dim3(128 * 64 / 256, 3), dim3(256), 0, st, job);
    }
    NarrowArgs g{x, W1, b1, b2, b3, b4, P2, P3, P4, z1, z2, z3, z4, R, zero_keys, zero_keys ? zero_n : 0};
    const size_t lds = (size_t)(2 * 3 * 64 * 72 + 3 * 128 * 72) * 2 + 64 * 4 * 4 + 4 * 32 * 68 * 4;
    static SnLdsAttr at, af;
    if (sn_lds_attr(at, (const void *)pointnet_narrow_fwd_kernel<true>, lds, "sn_pointnet_narrow_forward") ||
        sn_lds_attr(af, (const void *)pointnet_narrow_fwd_kernel<false>, lds, "sn_pointnet_narrow_forward"))
        return SN_ERR_UNSUPPORTED;
    const dim3 grid((R + kNarrowRows - 1) / kNarrowRows);
    if (z1)
        hipLaunchKernelGGL(pointnet_narrow_fwd_kernel<true>, grid, dim3(256), lds, st, g);
    else
        hipLaunchKernelGGL(pointnet_narrow_fwd_kernel<false>, grid, dim3(256), lds, st, g);
    SN_LAUNCH_CHECK();
    return 0;
}

// ---- the data gradient back through that narrow front in ONE launch (frozen weights: no weight gradients wanted): from dL/dz4
// (R,128) -- what the pooled layer's backward hands down -- to the gradient of the cloud (R,3):
//   dz3 = [z3 > 0] (dz4 W4),  dz2 = [z2 > 0] (dz3 W3),  dz1 = [z1 > 0] (dz2 W2),  dx = dz1 W1
// the mirror of pointnet_narrow_fwd_kernel: a wave takes 32 rows, the weights arrive as TRANSPOSED bf16 planes ([3][Ci][Co]: the
// B fragment of a data-gradient product is 8 consecutive output channels of one input channel), each product's 32 x 64 tile goes
// through the wave's LDS tile into the row layout, where the ReLU mask of the layer below is applied from a 16-byte read of its
// saved pre-activations; the 64 -> 3 product is 96 FMAs per lane and one cross-half add.  (4 launches of 5-7 us each before.)
struct NarrowBwdArgs {
    const float *dz4, *z1, *z2, *z3, *W1;
    const __bf16 *Q4, *Q3, *Q2;  // transposed planes [3][64][128], [3][64][64], [3][64][64]
    float *dx;
    int R;
};
struct SplitJobT3 {
    const float *w[3];
    __bf16 *dst[3];
    int co[3], ci[3];
};
__global__ void __launch_bounds__(256) split_planes_t3_kernel(SplitJobT3 job)
{
    const int l = blockIdx.y, co = job.co[l], ci = job.ci[l], n = co * ci;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;  // index into the TRANSPOSED image: i = c_in * co + c_out
    if (i >= n) return;
    const int c_in = i / co, c_out = i % co;
    __bf16 h1, h2, h3;
    split3(job.w[l][(size_t)c_out * ci + c_in], h1, h2, h3);
    job.dst[l][i] = h1, job.dst[l][(size_t)n + i] = h2, job.dst[l][2 * (size_t)n + i] = h3;
}
__global__ void __launch_bounds__(256) pointnet_narrow_bwd_kernel(NarrowBwdArgs g)
{
    constexpr int P4 = 136, P3 = 72, PT = 68;  // plane row pitches (bf16) for K = 128 / 64, transpose tile pitch (floats)
    constexpr int N4 = 3 * 64 * P4, N3 = 3 * 64 * P3;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __bf16 *S4 = reinterpret_cast<__bf16 *>(lds), *S3 = S4 + N4, *S2 = S3 + N3;
    float *W1s = reinterpret_cast<float *>(S2 + N3);  // [64][4] = (w0, w1, w2, -)
    float *Tall = W1s + 64 * 4;
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float *T = Tall + wave * (32 * PT);
    const int row0 = blockIdx.x * kNarrowRows + wave * 32;
    const int rrow = min(row0 + l31, g.R - 1);
    const bool rok = row0 + l31 < g.R;
    {
        // every load of the prologue in flight at once: the planes (12 + 6 + 6 items of 16 bytes per thread) and this lane's 128
        // gradient values (its row's k-groups)
        bf16x8 r4[12], r3[6], r2[6];
#pragma unroll
        for (int q = 0; q < 12; ++q) r4[q] = *reinterpret_cast<const bf16x8 *>(g.Q4 + (size_t)(tid + q * 256) * 8);
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            r3[q] = *reinterpret_cast<const bf16x8 *>(g.Q3 + (size_t)(tid + q * 256) * 8);
            r2[q] = *reinterpret_cast<const bf16x8 *>(g.Q2 + (size_t)(tid + q * 256) * 8);
        }
#pragma unroll
        for (int q = 0; q < 12; ++q) {
            const int f = tid + q * 256, row = f >> 4, k8 = (f & 15) * 8;  // row = plane * 64 + input channel; global rows are 128 wide
            *reinterpret_cast<bf16x8 *>(S4 + row * P4 + k8) = r4[q];
        }
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            const int f = tid + q * 256, row = f >> 3, k8 = (f & 7) * 8;
            *reinterpret_cast<bf16x8 *>(S3 + row * P3 + k8) = r3[q];
            *reinterpret_cast<bf16x8 *>(S2 + row * P3 + k8) = r2[q];
        }
        if (tid < 64) W1s[tid * 4 + 0] = g.W1[tid * 3 + 0], W1s[tid * 4 + 1] = g.W1[tid * 3 + 1], W1s[tid * 4 + 2] = g.W1[tid * 3 + 2], W1s[tid * 4 + 3] = 0.f;
    }
    // dz4 -> A fragments (K = 128: eight k-steps)
    bf16x8 a8[3][8];
    {
        const float *ar = g.dz4 + (size_t)rrow * 128 + 8 * h;
        float4 v[8][2];
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) v[kk][0] = *reinterpret_cast<const float4 *>(ar + kk * 16), v[kk][1] = *reinterpret_cast<const float4 *>(ar + kk * 16 + 4);
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            const float e[8] = {v[kk][0].x, v[kk][0].y, v[kk][0].z, v[kk][0].w, v[kk][1].x, v[kk][1].y, v[kk][1].z, v[kk][1].w};
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                __bf16 h1, h2, h3;
                split3(rok ? e[t] : 0.f, h1, h2, h3);
                a8[0][kk][t] = h1, a8[1][kk][t] = h2, a8[2][kk][t] = h3;
            }
        }
    }
    __syncthreads();
    // acc (two 32-column tiles) -> T -> this lane's row values d[kk][8], masked by the saved pre-activations zmask (R, 64)
    float d[4][8];
    const auto finish = [&](f32x16 (&acc)[2], const float *zmask) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) T[frag_row(e, lane) * PT + j * 32 + l31] = acc[j][e];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const float4 v0 = *reinterpret_cast<const float4 *>(T + l31 * PT + kk * 16 + 8 * h), v1 = *reinterpret_cast<const float4 *>(T + l31 * PT + kk * 16 + 8 * h + 4);
            const float *zp = zmask + (size_t)rrow * 64 + kk * 16 + 8 * h;
            const float4 z0 = *reinterpret_cast<const float4 *>(zp), z1 = *reinterpret_cast<const float4 *>(zp + 4);
            const float e[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w}, zz[8] = {z0.x, z0.y, z0.z, z0.w, z1.x, z1.y, z1.z, z1.w};
#pragma unroll
            for (int t = 0; t < 8; ++t) d[kk][t] = zz[t] > 0.f ? e[t] : 0.f;
        }
        __builtin_amdgcn_wave_barrier();
    };
    bf16x8 a[3][4];
    const auto refrag = [&] {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                __bf16 h1, h2, h3;
                split3(d[kk][t], h1, h2, h3);
                a[0][kk][t] = h1, a[1][kk][t] = h2, a[2][kk][t] = h3;
            }
    };
#define SN_NB_TERM(A, B, PA, PB) _Pragma("unroll") for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[PA][kk], B[PB][j], acc[j], 0, 0, 0)
    {   // dz3 = [z3 > 0] (dz4 . W4): K = 128
        f32x16 acc[2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
        bf16x8 b[2][3][2];
        const auto load_b = [&](int kk, bf16x8 (&bb)[3][2]) {
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int j = 0; j < 2; ++j) bb[p][j] = *reinterpret_cast<const bf16x8 *>(S4 + (p * 64 + j * 32 + l31) * P4 + kk * 16 + 8 * h);
        };
        load_b(0, b[0]);
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            if (kk + 1 < 8) load_b(kk + 1, b[(kk + 1) & 1]);
            SN_NB_TERM(a8, b[kk & 1], 0, 2);
            SN_NB_TERM(a8, b[kk & 1], 2, 0);
            SN_NB_TERM(a8, b[kk & 1], 1, 1);
            SN_NB_TERM(a8, b[kk & 1], 0, 1);
            SN_NB_TERM(a8, b[kk & 1], 1, 0);
            SN_NB_TERM(a8, b[kk & 1], 0, 0);
        }
        finish(acc, g.z3);
    }
    const auto layer64 = [&](const __bf16 *S, const float *zmask) {
        refrag();
        f32x16 acc[2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
        bf16x8 b[4][3][2];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int j = 0; j < 2; ++j) b[kk][p][j] = *reinterpret_cast<const bf16x8 *>(S + (p * 64 + j * 32 + l31) * P3 + kk * 16 + 8 * h);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            SN_NB_TERM(a, b[kk], 0, 2);
            SN_NB_TERM(a, b[kk], 2, 0);
            SN_NB_TERM(a, b[kk], 1, 1);
            SN_NB_TERM(a, b[kk], 0, 1);
            SN_NB_TERM(a, b[kk], 1, 0);
            SN_NB_TERM(a, b[kk], 0, 0);
        }
        finish(acc, zmask);
    };
#undef SN_NB_TERM
    layer64(S3, g.z2);  // dz2 = [z2 > 0] (dz3 . W3)
    layer64(S2, g.z1);  // dz1 = [z1 > 0] (dz2 . W2)
    // dx = dz1 . W1: this lane's 32 channels, then the other half of the row
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const float4 w = *reinterpret_cast<const float4 *>(W1s + (kk * 16 + 8 * h + t) * 4);
            s0 = fmaf(d[kk][t], w.x, s0), s1 = fmaf(d[kk][t], w.y, s1), s2 = fmaf(d[kk][t], w.z, s2);
        }
    s0 += __shfl_xor(s0, 32), s1 += __shfl_xor(s1, 32), s2 += __shfl_xor(s2, 32);
    if (h == 0 && rok) {
        float *o = g.dx + (size_t)(row0 + l31) * 3;
        o[0] = s0, o[1] = s1, o[2] = s2;
    }
}

extern "C" int sn_pointnet_narrow_backward_supported(int R, int c1, int c2, int c3, int c4)
{
    return R >= 1 && c1 == 64 && c2 == 64 && c3 == 64 && c4 == 128;
}
// dx (R,3) from dz4 (R,128) and the saved pre-activations z1..z3 (R,64).  wplanes_t: 3 * 16384 bf16 for the TRANSPOSED split weights
// (planes_ready != 0: already holds them).
extern "C" int sn_pointnet_narrow_backward(int R, const float *dz4, const float *z1, const float *z2, const float *z3, const float *W1,
                                           const float *W2, const float *W3, const float *W4, void *wplanes_t, int planes_ready, float *dx,
                                           sn_stream_t stream)
{
    SN_REQUIRE(R >= 1 && dz4 && z1 && z2 && z3 && W1 && W2 && W3 && W4 && wplanes_t && dx, "bad argument");
    hipStream_t st = (hipStream_t)stream;
    __bf16 *Q4 = (__bf16 *)wplanes_t, *Q3 = Q4 + 3 * 128 * 64, *Q2 = Q3 + 3 * 64 * 64;
    if (!planes_ready) {
        SplitJobT3 job{{W4, W3, W2}, {Q4, Q3, Q2}, {128, 64, 64}, {64, 64, 64}};
        hipLaunchKernelGGL(split_planes_t3_kernel, dim3(128 * 64 / 256, 3), dim3(256), 0, st, job);
    }
    NarrowBwdArgs g{dz4, z1, z2, z3, W1, Q4, Q3, Q2, dx, R};
    const size_t lds = (size_t)(3 * 64 * 136 + 2 * 3 * 64 * 72) * 2 + 64 * 4 * 4 + 4 * 32 * 68 * 4;
    static SnLdsAttr attr;
    if (sn_lds_attr(attr, (const void *)pointnet_narrow_bwd_kernel, lds, "sn_pointnet_narrow_backward")) return SN_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(pointnet_narrow_bwd_kernel, dim3((R + kNarrowRows - 1) / kNarrowRows), dim3(256), lds, st, g);
    SN_LAUNCH_CHECK();
    return 0;
}

// ---- Linear layers on at most 32 rows (PCRNet's trunk, registration/models/pcrnet.py:56-77: 2048 -> 1024 -> 1024 -> 512 -> 512 ->
// 256 -> 7 on the batch's 32 feature vectors; forward and data gradient).  15.5 MB of weights against 1 MFLOP per row: the layer is
// a weight STREAM, and a CU pulls only ~25 GB/s from memory -- so the product is cut into (32-column tile) x (K slice) workgroups
// until the grid covers the chip (fc1: 32 x 8), four waves per workgroup each taking a quarter of the slice with ALL of its loads
// issued up front (1, 2 or 4 k-steps of 16), fp32 products as six bf16 MFMAs of three-way split operands (split in registers:
// the fragments are 8 consecutive k per lane, i.e. two 16-byte loads straight from the row-major operands, no LDS staging).
// The four waves' partial tiles are summed through LDS (wave_reduce_scatter4), the S slices by the workgroup that arrives last
// at the tile's counter (partials cross as write-through stores / sc1 loads, relaxed counter: see the FC chains) in slice order:
// deterministic.  Epilogue: bias, ReLU.  out (R, N) = act((x . [gate > 0]) (R, K) . W^T + bias):
//   wmode 0: W is (N, K) row-major -- forward, y = x W^T + b
//   wmode 1: W is (K, N) row-major -- data gradient, dX = (dY . [y > 0]) W with gate = the layer's own (post-ReLU) output
struct SkinnyArgs {
    const float *x, *gate, *W, *bias;
    float *out, *part;
    unsigned *counter;
    int R, K, N, S, kslice, wmode, relu;
    // two-part operands (the trunk's first layer reads the two clouds' feature vectors where they lie, its data gradient hands
    // each cloud its own gradient -- no concatenation / slice copies around the trunk):
    const float *x2;  // columns k >= ksplit of the input come from x2 (R, K - ksplit); x is then (R, ksplit).  NULL: x is (R, K)
    float *out2;      // columns n >= nsplit of the output go to out2 (R, N - nsplit); out is then (R, nsplit).  Either may be NULL
    int ksplit, nsplit;
};
// RT: 32-row tiles per workgroup (R <= 32 RT): the weight fragments -- the traffic that bounds the layer -- are loaded and split once
// and multiply every row tile (several task-network evaluations of one step batched into one trunk pass: PCRNet on the progressive
// sampler's prefixes).
template <int KSTEPS, int RT>
__global__ void __launch_bounds__(256) skinny_linear_kernel(SkinnyArgs g)
{
    __shared__ __attribute__((aligned(16))) float red[kRsFloats];
    __shared__ int s_last;
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile = blockIdx.x, s = blockIdx.y, n0 = tile * 32;
    const int K = g.K, N = g.N, R = g.R;
    const int kw = g.kslice / 4;                                  // this wave's K range: KSTEPS steps of 16
    const int kb = s * g.kslice + wave * kw;
    const int n = n0 + l31;
    const bool nok = n < N;
    const bool kvec = (K & 3) == 0;
    SN_TL(0);
    float ea[RT][KSTEPS][8], eg[RT][KSTEPS][8], eb[KSTEPS][8];
#pragma unroll
    for (int st = 0; st < KSTEPS; ++st) {
        const int k8 = kb + st * 16 + 8 * h;
        const bool full = k8 + 8 <= K && kvec;
        // A: 8 consecutive k of row m of x (and of the gate), for every row tile
        const float *xs = g.x;
        int ldx = K, kx = k8;
        if (g.x2) {  // (ksplit is a multiple of 8: a fragment never straddles the two parts)
            if (k8 >= g.ksplit) xs = g.x2, ldx = K - g.ksplit, kx = k8 - g.ksplit;
            else ldx = g.ksplit;
        }
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const int m = rt * 32 + l31;
            const bool mok = m < R;
            if (full && mok) {
                const float4 v0 = *reinterpret_cast<const float4 *>(xs + (size_t)m * ldx + kx), v1 = *reinterpret_cast<const float4 *>(xs + (size_t)m * ldx + kx + 4);
                ea[rt][st][0] = v0.x, ea[rt][st][1] = v0.y, ea[rt][st][2] = v0.z, ea[rt][st][3] = v0.w;
                ea[rt][st][4] = v1.x, ea[rt][st][5] = v1.y, ea[rt][st][6] = v1.z, ea[rt][st][7] = v1.w;
                if (g.gate) {
                    const float4 g0 = *reinterpret_cast<const float4 *>(g.gate + (size_t)m * K + k8), g1 = *reinterpret_cast<const float4 *>(g.gate + (size_t)m * K + k8 + 4);
                    eg[rt][st][0] = g0.x, eg[rt][st][1] = g0.y, eg[rt][st][2] = g0.z, eg[rt][st][3] = g0.w;
                    eg[rt][st][4] = g1.x, eg[rt][st][5] = g1.y, eg[rt][st][6] = g1.z, eg[rt][st][7] = g1.w;
                }
            } else {
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const bool ok = mok && k8 + t < K;
                    ea[rt][st][t] = ok ? xs[(size_t)m * ldx + kx + t] : 0.f;
                    if (g.gate) eg[rt][st][t] = ok ? g.gate[(size_t)m * K + k8 + t] : 0.f;
                }
            }
        }
        // B: 8 consecutive k of output column n
        if (g.wmode == 0) {
            if (full && nok) {
                const float4 v0 = *reinterpret_cast<const float4 *>(g.W + (size_t)n * K + k8), v1 = *reinterpret_cast<const float4 *>(g.W + (size_t)n * K + k8 + 4);
                eb[st][0] = v0.x, eb[st][1] = v0.y, eb[st][2] = v0.z, eb[st][3] = v0.w, eb[st][4] = v1.x, eb[st][5] = v1.y, eb[st][6] = v1.z, eb[st][7] = v1.w;
            } else {
#pragma unroll
                for (int t = 0; t < 8; ++t) eb[st][t] = (nok && k8 + t < K) ? g.W[(size_t)n * K + k8 + t] : 0.f;
            }
        } else {
#pragma unroll
            for (int t = 0; t < 8; ++t) eb[st][t] = (nok && k8 + t < K) ? g.W[(size_t)(k8 + t) * N + n] : 0.f;  // (lanes: consecutive n)
        }
    }
    f32x16 acc[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[rt][e] = 0.f;
#pragma unroll
    for (int st = 0; st < KSTEPS; ++st) {
        bf16x8 b[3];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            __bf16 h1, h2, h3;
            split3(eb[st][t], h1, h2, h3);
            b[0][t] = h1, b[1][t] = h2, b[2][t] = h3;
        }
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            bf16x8 a[3];
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const float av = g.gate ? (eg[rt][st][t] > 0.f ? ea[rt][st][t] : 0.f) : ea[rt][st][t];
                __bf16 h1, h2, h3;
                split3(av, h1, h2, h3);
                a[0][t] = h1, a[1][t] = h2, a[2][t] = h3;
            }
#define SN_SK_TERM(PA, PB) acc[rt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA], b[PB], acc[rt], 0, 0, 0)
            SN_SK_TERM(0, 2);
            SN_SK_TERM(2, 0);
            SN_SK_TERM(1, 1);
            SN_SK_TERM(0, 1);
            SN_SK_TERM(1, 0);
            SN_SK_TERM(0, 0);
#undef SN_SK_TERM
        }
    }
    SN_TL(1);
    // wave w now holds column 8 w + (lane >> 3) of each row tile, rows 4 (lane & 7) .. + 3
    float4 v[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        if (rt > 0) lds_barrier();  // (the exchange area is read by other waves until they pass this point)
        v[rt] = wave_reduce_scatter4(acc[rt], red);
    }
    SN_TL(2);
    const int S = g.S;
    typedef float sk4 __attribute__((ext_vector_type(4)));
    if (S > 1) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            float *P = g.part + (((size_t)s * gridDim.x + tile) * RT + rt) * 1024 + (size_t)tid * 4;
            const sk4 pv = {v[rt].x, v[rt].y, v[rt].z, v[rt].w};
            asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 4" ::"v"(P), "v"(pv) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        SN_TL(3);
        if (tid == 0) {
            const unsigned t = __hip_atomic_fetch_add(g.counter + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_last = t == (unsigned)(S - 1);
            if (s_last) __hip_atomic_store(g.counter + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // armed for the next launch
        }
        __syncthreads();
        SN_TL(4);
        if (!s_last) return;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            sk4 accv = {0.f, 0.f, 0.f, 0.f};
            for (int q0 = 0; q0 < S; q0 += 8) {  // slices in ascending order, eight loads in flight
                sk4 r[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int qq = min(q0 + q, S - 1);
                    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=&v"(r[q]) : "v"(g.part + (((size_t)qq * gridDim.x + tile) * RT + rt) * 1024 + (size_t)tid * 4) : "memory");
                }
                // (the loaded registers are operands of the wait: register-only uses of them must not be scheduled above it)
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7])::"memory");
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    if (q0 + q < S) accv = accv + r[q];
            }
            v[rt] = make_float4(accv.x, accv.y, accv.z, accv.w);
        }
        SN_TL(5);
    }
    const int col = n0 + wave * 8 + (lane >> 3), r0 = 4 * (lane & 7);
    if (col < N) {
        const float bv = g.bias ? g.bias[col] : 0.f;
        float *dst = g.out;
        int ldo = N, c = col;
        if (g.nsplit > 0) {
            if (col >= g.nsplit) dst = g.out2, ldo = N - g.nsplit, c = col - g.nsplit;
            else ldo = g.nsplit;
        }
        if (dst)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                const float o[4] = {v[rt].x + bv, v[rt].y + bv, v[rt].z + bv, v[rt].w + bv};
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (rt * 32 + r0 + i < R) dst[(size_t)(rt * 32 + r0 + i) * ldo + c] = g.relu ? relu_np(o[i]) : o[i];
            }
    }
}

// S (K slices) and k-steps per wave for a (K, N) layer.  A slice is a multiple of 64 (one k-step of 16 for each of the four waves);
// start from one k-step per wave (S = K / 64 slices: the most workgroups) and double the k-steps, halving S, while the grid stays
// at 512 workgroups or more -- two per CU is where more of them stop buying memory parallelism -- or there are more than 8 slices;
// at most 4 k-steps a wave (the kernel keeps all of a wave's loads in flight).
static void skinny_plan(int K, int N, int &S, int &ksteps)
{
    const int tiles = (N + 31) / 32;
    const int k64 = (K + 63) / 64;
    S = k64, ksteps = 1;
    while (S % 2 == 0 && ksteps < 4 && (tiles * S >= 512 || S > 8)) S /= 2, ksteps *= 2;
    // (S > 8: the last workgroup of a tile sums the slices from one batch of eight loads in flight; a second batch is a second
    // memory round trip -- tools/skinny_timeline.py: 1024 -> 512 with 16 slices spent 3.1 us there, 1.6 with 8)
}
extern "C" int sn_skinny_linear_supported(int R, int K, int N)
{
    if (R < 1 || R > 128 || K < 1 || N < 1) return 0;
    int S, ks;
    skinny_plan(K, N, S, ks);
    return ks <= 4;
}
extern "C" long long sn_skinny_linear_scratch_bytes(int R, int K, int N)
{
    int S, ks;
    skinny_plan(K, N, S, ks);
    const int rt = R <= 32 ? 1 : R <= 64 ? 2 : 4;
    return (long long)S * ((N + 31) / 32) * rt * 1024 * (long long)sizeof(float);
}
// counters: (N + 31) / 32 zeroed 32-bit words (left zeroed).  transposed != 0: W is (K, N) (the data gradient through a layer
// whose weight is (Co = K, Ci = N)).
extern "C" int sn_skinny_linear2(int R, int K, int N, const float *x, const float *x2, int ksplit, const float *gate, const float *W,
                                 int transposed, const float *bias, int relu, float *out, float *out2, int nsplit, float *scratch,
                                 unsigned *counters, sn_stream_t stream);
extern "C" int sn_skinny_linear(int R, int K, int N, const float *x, const float *gate, const float *W, int transposed, const float *bias,
                                int relu, float *out, float *scratch, unsigned *counters, sn_stream_t stream)
{
    SN_REQUIRE(out, "null pointer");
    return sn_skinny_linear2(R, K, N, x, nullptr, 0, gate, W, transposed, bias, relu, out, nullptr, 0, scratch, counters, stream);
}
// The two-part form: x2 / ksplit -- input columns k >= ksplit come from x2 (R, K - ksplit), x is (R, ksplit) (x2 == NULL: x is (R, K));
// out2 / nsplit -- output columns n >= nsplit go to out2 (R, N - nsplit), out is (R, nsplit) (nsplit == 0: out is (R, N)); with
// nsplit > 0 either output may be NULL (that part is not wanted).
extern "C" int sn_skinny_linear2(int R, int K, int N, const float *x, const float *x2, int ksplit, const float *gate, const float *W,
                                 int transposed, const float *bias, int relu, float *out, float *out2, int nsplit, float *scratch,
                                 unsigned *counters, sn_stream_t stream)
{
    SN_REQUIRE(x && W && scratch && counters, "null pointer");
    SN_REQUIRE(!x2 || (ksplit > 0 && ksplit < K && ksplit % 8 == 0 && !gate), "x2: ksplit must be a multiple of 8 inside (0, K), no gate");
    SN_REQUIRE(nsplit >= 0 && nsplit < N && (nsplit > 0 ? (out || out2) : out != nullptr), "bad output split");
    if (!sn_skinny_linear_supported(R, K, N)) return sn_set_error(SN_ERR_UNSUPPORTED, "sn_skinny_linear: needs at most 128 rows");
    SkinnyArgs g{};
    g.x = x, g.gate = gate, g.W = W, g.bias = bias, g.out = out, g.part = scratch, g.counter = counters;
    g.R = R, g.K = K, g.N = N, g.wmode = transposed ? 1 : 0, g.relu = relu;
    g.x2 = x2, g.ksplit = x2 ? ksplit : 0, g.out2 = out2, g.nsplit = nsplit;
    int ks;
    skinny_plan(K, N, g.S, ks);
    g.kslice = ks * 64;
    const dim3 grid((N + 31) / 32, g.S);
    hipStream_t st = (hipStream_t)stream;
#define SN_SK_LAUNCH(RT_)                                                                                 \
    do {                                                                                                  \
        if (ks == 1) hipLaunchKernelGGL((skinny_linear_kernel<1, RT_>), grid, dim3(256), 0, st, g);       \
        else if (ks == 2) hipLaunchKernelGGL((skinny_linear_kernel<2, RT_>), grid, dim3(256), 0, st, g);  \
        else hipLaunchKernelGGL((skinny_linear_kernel<4, RT_>), grid, dim3(256), 0, st, g);               \
    } while (0)
    if (R <= 32) SN_SK_LAUNCH(1);
    else if (R <= 64) SN_SK_LAUNCH(2);
    else SN_SK_LAUNCH(4);
#undef SN_SK_LAUNCH
    SN_LAUNCH_CHECK();
    return 0;
}

// ---- weight gradient of those layers (round 4: a TRAINABLE trunk stays on the library -- registration/main.py --train-pcrnet,
// models/pcrnet.py:62-82): dW (N, K) = dZ^T . X with dZ = dY . [gate > 0] (R, N) and X = [x | x2] (R, K); db (N) = column sums of
// dZ.  R <= 128 rows are the whole contraction: a wave owns one 32 x 32 tile of dW and runs R / 2 fp32 MFMAs (32x32x2: exact
// products, rows in ascending order -> deterministic), operands straight from memory (a row of dZ / X per lane pair, 128-byte
// segments); the output -- 8 MB for PCRNet's first layer -- is the traffic.  Four waves per workgroup = 32 rows x 128 columns of dW.
__global__ void __launch_bounds__(256) skinny_wgrad_kernel(int R, int K, int N, const float *__restrict__ x, const float *__restrict__ x2,
                                                           int ksplit, const float *__restrict__ dy, const float *__restrict__ gate,
                                                           float *__restrict__ dW, float *__restrict__ db)
{
    const int lane = threadIdx.x & 63, l31 = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n0 = blockIdx.y * 32, k0 = (blockIdx.x * 4 + wave) * 32;
    if (k0 >= K) return;
    const int n = n0 + l31, k = k0 + l31;
    const bool nok = n < N, kok = k < K;
    // column k of X: from x (R, ksplit) or x2 (R, K - ksplit)
    const float *xs = x;
    int xk = k, xld = K;
    if (x2) {
        if (k < ksplit) xld = ksplit;
        else xs = x2, xk = k - ksplit, xld = K - ksplit;
    }
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    float bsum = 0.f;
    for (int r0 = 0; r0 < R; r0 += 2) {
        const int r = r0 + h;
        float a = 0.f, b = 0.f;
        if (r < R) {
            if (nok) {
                a = dy[(size_t)r * N + n];
                if (gate) a = gate[(size_t)r * N + n] > 0.f ? a : 0.f;
            }
            if (kok) b = xs[(size_t)r * xld + xk];
        }
        bsum += a;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int nn = n0 + frag_row(e, lane);
        if (nn < N && kok) dW[(size_t)nn * K + k] = acc[e];
    }
    if (db && k0 == 0) {  // (the k-block 0 wave of every row block: even rows in lanes 0..31, odd rows in 32..63)
        bsum += __shfl_xor(bsum, 32);
        if (lane < 32 && nok) db[n] = bsum;
    }
}

extern "C" int sn_skinny_wgrad(int R, int K, int N, const float *x, const float *x2, int ksplit, const float *dy, const float *gate,
                               float *dW, float *db, sn_stream_t stream)
{
    SN_REQUIRE(R >= 1 && R <= 256 && K >= 1 && N >= 1, "bad size (at most 256 rows)");
    SN_REQUIRE(x && dy && dW, "null pointer");
    SN_REQUIRE(!x2 || (ksplit > 0 && ksplit < K), "x2: ksplit inside (0, K)");
    hipLaunchKernelGGL(skinny_wgrad_kernel, dim3((K + 127) / 128, (N + 31) / 32), dim3(256), 0, (hipStream_t)stream, R, K, N, x, x2,
                       x2 ? ksplit : 0, dy, gate, dW, db);
    SN_LAUNCH_CHECK();
    return 0;
}

// ---- data gradient of that last layer when its dZ is SPARSE (BatchNorm-free stack: dZ = dY through the max over the points has one
// non-zero per cloud and channel -- the pooled element; with a BatchNorm the k2 Z + k3 terms make it dense and the GEMM kernels
// apply).  dYprev[b, n, :] = relu'_prev . sum over the channels c whose maximum sits at point n of  gsel[b][c] * W[c][:],
// gsel = pooled > 0 ? g : 0 (the pooling backward, folded in).  One workgroup per (cloud, 32 input channels, 64 rows): 16 groups of 32
// lanes (two per wave); group q adds its channels (c = q, q + 16, ...: ascending) into its own copy of the cloud's (npts <= 64) x 32
// tile in LDS -- a lane is the only writer of its column of its copy, so the read-modify-writes need no atomics and their order
// is fixed; the 16 copies are summed in group order: deterministic, no workgroup talks to another.  The (row, gradient) pairs of
// 32 channels sit one per lane and reach the half-waves as scalars (v_readlane) + one select; a wave's region is [64][2][32]
// floats, so a lane's bank is its lane id whatever the rows are.  32 clouds x 1024 channels: 33 k rank-1 updates of 128 floats
// instead of the dense (2048 x 1024) x (1024 x 128) GEMM with its 64 workgroups of 32 dependent K chunks (DESIGN 5a').
constexpr int kPdsThreads = 512, kPdsGroups = 16, kPdsCi = 32, kPdsPts = 64, kPdsCh = 32;
__global__ void __launch_bounds__(kPdsThreads) pool_dgrad_sparse_kernel(int npts, int Ci, int Co, const float *__restrict__ g,
                                                                        const float *__restrict__ pooled, const int *__restrict__ argsel,
                                                                        const float *__restrict__ W, const float *__restrict__ zprev,
                                                                        const float *__restrict__ coef_prev, float *__restrict__ dyprev)
{
    __shared__ __attribute__((aligned(16))) float lds[8 * kPdsPts * 64];  // [wave][row][half][32]
    const int b = blockIdx.x, ci0 = blockIdx.y * kPdsCi;
    const int r0 = blockIdx.z * kPdsPts, nch = min(kPdsPts, npts - r0);  // this workgroup's rows of the cloud (clouds of up to 256 points)
    const int lane = threadIdx.x & 63, l = lane & 31, hh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int grp = wave * 2 + hh;
    const int ci = ci0 + l;
    const bool act = ci < Ci;
    float *out = lds + wave * (kPdsPts * 64) + hh * 32 + l;  // + row * 64
    // A workgroup whose rows hold no maximum of its cloud writes zeros and leaves before it has touched the weights or its LDS tile:
    // clouds that are cyclic repetitions of a shorter cloud (ops.cyclic_pad_cat: the progressive sampler's prefixes as one batch) have
    // every maximum in their first copy, i.e. three of the four row chunks of a 64-point cloud padded to 256 are empty.
    if (gridDim.z > 1) {  // (a cloud of one chunk holds all of its maxima: nothing to ask)
        bool hit = false;
        for (int c = threadIdx.x; c < Co; c += kPdsThreads) hit |= (unsigned)(argsel[(size_t)b * Co + c] - r0) < (unsigned)nch;
        if (!__syncthreads_or(hit)) {
            for (int e = threadIdx.x; e < nch * kPdsCi; e += kPdsThreads) {
                const int n = e >> 5, col = e & 31;
                if (ci0 + col < Ci) dyprev[((size_t)b * npts + r0 + n) * Ci + ci0 + col] = 0.f;
            }
            return;
        }
    }
    {
        float4 *z4 = reinterpret_cast<float4 *>(lds);
        for (int i = threadIdx.x; i < 8 * kPdsPts * 64 / 4; i += kPdsThreads) z4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();  // (another wave zeroed part of this wave's region)
    const size_t bo = (size_t)b * Co;
    for (int k0 = 0; k0 * kPdsGroups < Co; k0 += kPdsCh) {  // 32 channels of every group per pass
        // every load of the pass leaves first: the lane's 32 weights, and the (row, gradient) pair of its group's l-th channel
        float w[kPdsCh];
#pragma unroll
        for (int k = 0; k < kPdsCh; ++k) {
            const int c = grp + kPdsGroups * (k0 + k);
            w[k] = (act && c < Co) ? W[(size_t)c * Ci + ci] : 0.f;
        }
        int nv = 0;
        float vv = 0.f;
        {
            const int c = grp + kPdsGroups * (k0 + l);
            if (c < Co) {
                const int nn = argsel[bo + c];
                const float gv = pooled[bo + c] > 0.f ? g[bo + c] : 0.f;  // (the pooling backward: sn_pool_backward's expression)
                const bool in = (unsigned)(nn - r0) < (unsigned)nch;
                nv = in ? nn - r0 : 0, vv = in ? gv : 0.f;
            }
        }
#pragma unroll
        for (int k = 0; k < kPdsCh; ++k) {  // ascending channels: the order of every column's sum is fixed
            const int na = __builtin_amdgcn_readlane(nv, k), nb = __builtin_amdgcn_readlane(nv, 32 + k);
            const float va = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(vv), k)),
                        vb = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(vv), 32 + k));
            if (va == 0.f && vb == 0.f) continue;  // (scalars: neither half-wave's channel has its maximum in this workgroup's rows)
            float *o = out + (hh ? nb : na) * 64;
            *o = fmaf(hh ? vb : va, w[k], *o);
        }
    }
    __syncthreads();
    const float *sc = coef_prev, *sh = coef_prev ? coef_prev + Ci : nullptr;
    for (int e = threadIdx.x; e < nch * kPdsCi; e += kPdsThreads) {
        const int n = e >> 5, col = e & 31;
        float a = 0.f;
#pragma unroll
        for (int q = 0; q < kPdsGroups; ++q) a += lds[(q >> 1) * (kPdsPts * 64) + n * 64 + (q & 1) * 32 + col];
        if (ci0 + col < Ci) {
            const size_t o = ((size_t)b * npts + r0 + n) * Ci + ci0 + col;
            if (zprev) a = fmaf(zprev[o], sc ? sc[ci0 + col] : 1.f, sh ? sh[ci0 + col] : 0.f) > 0.f ? a : 0.f;
            dyprev[o] = a;
        }
    }
}

extern "C" int sn_pool_dgrad_sparse_supported(int B, int npts, int Ci, int Co)
{
    return B >= 1 && npts >= 1 && npts <= 4 * kPdsPts && Ci >= 1 && Co >= 1;  // (beyond 256 points the dense kernels are ahead)
}
extern "C" int sn_pool_dgrad_sparse(int B, int npts, int Ci, int Co, const float *g, const float *pooled, const int *argsel,
                                    const float *W, const float *zprev, const float *coef_prev, float *dyprev, sn_stream_t stream)
{
    SN_REQUIRE(g && pooled && argsel && W && dyprev, "null pointer");
    if (!sn_pool_dgrad_sparse_supported(B, npts, Ci, Co))
        return sn_set_error(SN_ERR_UNSUPPORTED, "sn_pool_dgrad_sparse: needs at most 256 points per cloud");
    hipLaunchKernelGGL(pool_dgrad_sparse_kernel, dim3(B, (Ci + kPdsCi - 1) / kPdsCi, (npts + kPdsPts - 1) / kPdsPts), dim3(kPdsThreads), 0, (hipStream_t)stream, npts, Ci,
                       Co, g, pooled, argsel, W, zprev, coef_prev, dyprev);
    SN_LAUNCH_CHECK();
    return 0;
}
