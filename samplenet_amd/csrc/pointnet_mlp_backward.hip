// pointnet_mlp_backward.hip -- the BACKWARD side of the PointNet feature extractor + FC head (see pointnet_mlp.hip for the layer
// algebra): data-gradient and weight-gradient GEMM tiles, the fused convolution backward (fp32 MFMA and split-bf16 variants:
// one persistent kernel per layer that reads dY and Z once for both gradients), the R <= 32 layer kernels, the xyz input layer,
// BatchNorm-backward coefficient / weight-gradient reduction kernels, max-pool backward, and the conv-stack backward entry points.
#include "mlp_host.h"

namespace sn {

// ------------------------------------------------------------------------------------------------
// dgrad:  dYprev[R][Ci] = relu_mask_prev . ( dZ[R][Co] . W[Co][Ci] ) ; stats partial [gridDim.x][2][Ci]
//         (sum dYprev, sum dYprev * Zprev).  prev.mode == ACT_NONE: plain store, no mask / stats.
// ------------------------------------------------------------------------------------------------
struct DgradArgs {
    DzSrc dz;
    WSrc w;
    ActSrc prev;  // pre-BN activations + BN coefficients of the previous layer (for the ReLU mask)
    float *dyprev;
    float *stats;
    BnBwd bb;  // small-R kernels only: BatchNorm backward coefficients of the previous layer in the epilogue
};

template <class T, bool FULL, int ZMODE, int PMODE>
__device__ __forceinline__ void dgrad_body(const DgradArgs &g, int bx, int by, float *lds)
{
    SN_TL(0);
    SN_TL_ID(2);
    const int row0 = bx * T::BM, col0 = by * T::BN;
    const int R = g.dz.rows, Co = g.w.co, Ci = g.w.ci;
    f32x16 acc[T::TM][T::TN];
#pragma unroll
    for (int i = 0; i < T::TM; ++i)
#pragma unroll
        for (int j = 0; j < T::TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    const DzSrc dz = g.dz;
    const WSrc w = g.w;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wave / T::WC, wc = wave % T::WC;
    constexpr bool masked = PMODE == ACT_BN_RELU;
    // epilogue inputs (previous layer's pre-BN activations at this lane's output elements, its BN scale / shift) are
    // fetched BEFORE the GEMM so that their latency hides under it
    float zpv[T::TM][T::TN][16], scv[T::TN], shv[T::TN];
    if (masked) {
#pragma unroll
        for (int j = 0; j < T::TN; ++j) {
            const int col = col0 + (wc * T::TN + j) * 32 + (lane & 31);
            const int cc = (FULL || col < Ci) ? col : 0;
            scv[j] = g.prev.scale[cc], shv[j] = g.prev.shift[cc];
#pragma unroll
            for (int i = 0; i < T::TM; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int row = row0 + (wr * T::TM + i) * 32 + frag_row(e, lane);
                    zpv[i][j][e] = g.prev.z[(FULL || (row < R && col < Ci)) ? (size_t)row * Ci + col : 0];
                }
        }
    }
    // A: dZ rows, k = co contiguous.  B[k = co][x = ci]: W row-major is exactly [K][X], x contiguous.
    gemm_tile<T, true, false>(
        acc, Co, [&](int x, int k) { return dz.template load_c4<FULL, ZMODE>(row0 + x, k); },
        [&](int x, int k) { return w.template load_ci4<FULL>(k, col0 + x); }, lds);

    float s0[T::TN], s1[T::TN];
#pragma unroll
    for (int j = 0; j < T::TN; ++j) {
        const int col = col0 + (wc * T::TN + j) * 32 + (lane & 31);
        const float sc = masked ? scv[j] : 0.f, sh = masked ? shv[j] : 0.f;
        s0[j] = 0.f, s1[j] = 0.f;
#pragma unroll
        for (int i = 0; i < T::TM; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = row0 + (wr * T::TM + i) * 32 + frag_row(e, lane);
                if (FULL || (row < R && col < Ci)) {
                    float v = acc[i][j][e];
                    if (masked) {
                        const float zp = zpv[i][j][e];
                        v = (fmaf(zp, sc, sh) > 0.f) ? v : 0.f;
                        s0[j] += v;
                        s1[j] += v * zp;
                    }
                    g.dyprev[(size_t)row * Ci + col] = v;
                }
            }
    }
    SN_TL(3);
    SN_TL_DRAIN();
    SN_TL(4);
    if (masked && g.stats) {
        float *st = g.stats + (size_t)bx * 2 * Ci;
        column_reduce2<T>(s0, s1, lds, st, st + Ci, col0, Ci);
    }
    SN_TL(5);
}

template <class T, bool FULL, int ZMODE, int PMODE>
__global__ void __launch_bounds__(T::THREADS) linear_dgrad_kernel(DgradArgs g)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    dgrad_body<T, FULL, ZMODE, PMODE>(g, blockIdx.x, blockIdx.y, lds);
}

// ------------------------------------------------------------------------------------------------
// wgrad:  part[split][Co][Ci+1] = sum over the split's rows of dZ[r][co] * [act(prev)[r][ci] | 1]
// ------------------------------------------------------------------------------------------------
struct WgradArgs {
    DzSrc dz;
    ActSrc prev;  // ch = Ci (storage stride); ones_col = Ci when the bias-gradient column is requested, else -1
    float *part;
    int rows_per_split;
    int ncols;  // Ci + 1 with the bias column, Ci without
};

template <class T, bool FULL, int ZMODE, int PMODE>
__device__ __forceinline__ void wgrad_body(const WgradArgs &g, int bx, int by, int bz, float *lds)
{
    SN_TL(0);
    SN_TL_ID(1);
    const int m0 = bx * T::BM, n0 = by * T::BN;
    const int Co = g.dz.ch, Ce = g.ncols;
    const int r0 = bz * g.rows_per_split;
    const int r1 = min(g.dz.rows, r0 + g.rows_per_split);
    f32x16 acc[T::TM][T::TN];
#pragma unroll
    for (int i = 0; i < T::TM; ++i)
#pragma unroll
        for (int j = 0; j < T::TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    DzSrc dz = g.dz;
    ActSrc pv = g.prev;
    dz.rows = r1;  // rows beyond the split read as zero
    pv.rows = r1;
    // both operands: source [K = r][X], x contiguous
    gemm_tile<T, false, false>(
        acc, max(0, r1 - r0), [&](int x, int k) { return dz.template load_c4<FULL, ZMODE>(r0 + k, m0 + x); },
        [&](int x, int k) { return pv.template load_c4<FULL, PMODE>(r0 + k, n0 + x); }, lds);

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wave / T::WC, wc = wave % T::WC;
    float *P = g.part + (size_t)bz * Co * Ce;
#pragma unroll
    for (int j = 0; j < T::TN; ++j) {
        const int col = n0 + (wc * T::TN + j) * 32 + (lane & 31);
#pragma unroll
        for (int i = 0; i < T::TM; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = m0 + (wr * T::TM + i) * 32 + frag_row(e, lane);
                if (FULL || (row < Co && col < Ce)) P[(size_t)row * Ce + col] = acc[i][j][e];
            }
    }
    SN_TL(3);
    SN_TL_DRAIN();
    SN_TL(5);
}

template <class T, bool FULL, int ZMODE, int PMODE>
__global__ void __launch_bounds__(T::THREADS) linear_wgrad_kernel(WgradArgs g)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    wgrad_body<T, FULL, ZMODE, PMODE>(g, blockIdx.x, blockIdx.y, blockIdx.z, lds);
}

// Backward of one layer as ONE launch: the weight-gradient workgroups (MFMA-heavy: K = 128 rows per split) and the
// data-gradient workgroups (memory-heavy: read dY, Z, Zprev, write dYprev) are resident side by side, so the two
// kinds of phases overlap on every CU instead of running as two lock-stepped kernels.  Workgroup ids
// [0, n_w) -> wgrad (dispatched first: the longer of the two), [n_w, n_w + n_d) -> dgrad.
template <class T, int ZMODE, int PMODE>
__global__ void __launch_bounds__(T::THREADS) linear_bwd_kernel(DgradArgs d, WgradArgs w, int n_w, int wgx, int wgy, int dgx)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int id = blockIdx.x;
    if (id < n_w) {
        wgrad_body<T, true, ZMODE, PMODE>(w, id % wgx, (id / wgx) % wgy, id / (wgx * wgy), lds);
    } else {
        const int e = id - n_w;
        dgrad_body<T, true, ZMODE, PMODE>(d, e % dgx, e / dgx, lds);
    }
}

// ------------------------------------------------------------------------------------------------
// Fused backward of a 1x1-convolution layer with 64 / 128 channels on either side (R = B*N rows >> channels).
// The separate dgrad / wgrad kernels above each stream dZ (= k1 dY + k2 Z + k3) and the previous layer's
// activations from HBM; here ONE persistent workgroup per CU walks over 64-row tiles and uses every tile for both
// products while it sits in LDS:
//     dYprev[64 x CI]  = relu'(.) . ( dZ[64 x CO] . W[CO x CI] )      W lives in registers (B fragments, loaded once)
//     dWpart[CO x CI] += dZ^T[CO x 64] . relu(bn(Zprev))[64 x CI]     accumulated in registers over all tiles
// so Z / dY / Zprev are read once and the ReLU mask / BatchNorm-backward sums of the layer below come from the same
// LDS tile.  dZ is stored row-major with an even, non-multiple-of-4 leading dimension: the dgrad A fragment (transposed
// read, lane = row) and the wgrad A fragment (lane = channel) are both bank-conflict-free.  Tiles are double-buffered
// in LDS; the next tile's global loads are in flight under the current tile's MFMAs.
// 8 waves: waves 0-3 own the four dgrad tiles of a row tile, waves 4-7 the wgrad tiles -- every SIMD hosts one wave of
// each kind with the same MFMA count, so one wave's fragment reads / epilogue overlap the other's matrix work.
// Outputs: dYprev, stats partial [gridDim.x][2][CI] (sum dYprev, sum dYprev * Zprev), dW partial [gridDim.x][CO][CI].
// ------------------------------------------------------------------------------------------------
struct ConvBwdArgs {
    DzSrc dz;  // rows, ch = CO
    const float *W;
    const float *zprev, *scale_prev, *shift_prev;
    float *dyprev, *stats, *part;
    int ntiles;
    const float *xin;  // IN3 only: (R,3) input of the layer below when that layer is the xyz input layer
    const float *w_in, *b_in;  // IN3 with zprev == NULL (RZ1): the xyz layer's weights (CI, 3) / bias (CI) or NULL -- Zprev is rebuilt
                               // from the cloud (Z1[r][c] = W_in[c] . x_r + b_in[c], conv_in3_fwd_kernel's expression) instead of read
    // fixed-point statistics chain of the backward (sn_conv_stack_backward), the mirror of the forward's:
    //  acc_in  (DZ_BN): sums (sum dY, sum dY Z) of THIS layer's BatchNorm, left by the kernel of the layer above; every
    //          workgroup derives k1..k3 from them in its prologue, workgroup 0 also stores dgamma / dbeta / dbias (bb_in)
    //          and clears zero_ptr (what the previous kernel consumed);
    //  acc_out: the sums for the BatchNorm of the layer below go there by integer atomics instead of to `stats`.
    const long long *acc_in;
    BnBwd bb_in;
    long long *acc_out;
    long long *zero_ptr;
    int zero_n;
    // a 256-channel side as two passes of the 128 x 128 kernel (conv_bwd_bx3_kernel's GZ / GP / GW / DM): where a pass's
    // weight-gradient partial and statistics lie inside the layer's [G][Co][Ci] / [G][2][Ci] blocks (0: the kernel's own CO CI / CI / CI),
    // and the first pass's raw data gradient the second one adds (DM == 2; may be dyprev itself)
    int part_wg_stride, part_ld, stats_ld;
    const float *dyacc;
};

// Global-memory access of the dgrad waves goes through raw buffer instructions: resource (SGPRs) + per-lane byte offset
// that never changes (VGPR) + the tile's byte offset (SGPR).  The other wave of the SIMD keeps the matrix pipe busy and
// VALU instructions of this wave only find an issue slot now and then: with flat addressing the 64-bit per-lane address
// arithmetic in front of ~30 memory instructions made the top of every iteration take 2 us.  Out-of-range rows need no
// special casing either: loads beyond num_records return 0, stores are dropped.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t sn_rsrc;
__device__ __forceinline__ sn_rsrc make_rsrc(const void *p, unsigned bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ float4 buf_load4(sn_rsrc r, unsigned voff, unsigned soff)
{
    const u32x4 x = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    return make_float4(__uint_as_float(x.x), __uint_as_float(x.y), __uint_as_float(x.z), __uint_as_float(x.w));
}
typedef unsigned u32x3 __attribute__((ext_vector_type(3)));
__device__ __forceinline__ float4 buf_load3(sn_rsrc r, unsigned voff, unsigned soff)  // (x, y, z, 0)
{
    const u32x3 x = __builtin_amdgcn_raw_buffer_load_b96(r, voff, soff, 0);
    return make_float4(__uint_as_float(x.x), __uint_as_float(x.y), __uint_as_float(x.z), 0.f);
}
__device__ __forceinline__ int4 buf_load4i(sn_rsrc r, unsigned voff, unsigned soff)
{
    const u32x4 x = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    return make_int4((int)x.x, (int)x.y, (int)x.z, (int)x.w);
}
__device__ __forceinline__ float buf_load1(sn_rsrc r, unsigned voff, unsigned soff)
{
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ __forceinline__ void buf_store4(const float4 &v, sn_rsrc r, unsigned voff, unsigned soff)
{
    u32x4 x;
    x.x = __float_as_uint(v.x), x.y = __float_as_uint(v.y), x.z = __float_as_uint(v.z), x.w = __float_as_uint(v.w);
    __builtin_amdgcn_raw_buffer_store_b128(x, r, voff, soff, 0);
}

struct CbfRsrc {
    sn_rsrc z, dy, zprev, dyprev, argsel, gsel, dyacc;
};

template <int CO, int CI, int TR, int ZMODE, int NZ4, int NP4, bool SKIP_P = false, int GZ = CO, int GP = CI>
__device__ __forceinline__ void cbf_issue_loads(const CbfRsrc &rs, int tile, int b, unsigned zvo, unsigned pvo, unsigned avo,
                                                float4 (&rz)[NZ4], float4 (&rdy)[NZ4], float4 (&rp)[NP4], int4 &rag,
                                                float4 &rgs)
{
    constexpr int ZSTEP = 256 / (CO / 4), PSTEP = 256 / (CI / 4);  // staged by the 256 threads of the dgrad waves
    const unsigned zso = (unsigned)tile * (TR * GZ * 4), pso = (unsigned)tile * (TR * GP * 4);  // (GZ / GP: global row strides)
    // (the per-q strides ride in the SCALAR offset: as per-lane offsets they cost a VGPR each -- the split-bf16 kernel spilled them,
    //  and a spilled address reloaded in front of a load waits for every request before it)
#pragma unroll
    for (int q = 0; q < NZ4; ++q) {
        rz[q] = buf_load4(rs.z, zvo, zso + q * (ZSTEP * GZ * 4));
        if (ZMODE == DZ_BN) rdy[q] = buf_load4(rs.dy, zvo, zso + q * (ZSTEP * GZ * 4));
    }
#pragma unroll
    for (int q = 0; q < (SKIP_P ? 0 : NP4); ++q) rp[q] = buf_load4(rs.zprev, pvo, pso + q * (PSTEP * GP * 4));
    if (ZMODE == DZ_POOL) {  // the host guarantees npts % 64 == 0: one cloud (b) per tile
        rag = buf_load4i(rs.argsel, avo, (unsigned)b * (CO * 4));
        rgs = buf_load4(rs.gsel, avo, (unsigned)b * (CO * 4));
    }
}

template <int CO, int CI, int TR, int ZMODE, bool FULLR, int NZ4, int NP4>
__device__ __forceinline__ void cbf_stage(const ConvBwdArgs &g, int tile, int n0, int tid, float *__restrict__ Zs,
                                          float *__restrict__ Ps,
                                          const float4 (&rz)[NZ4], const float4 (&rdy)[NZ4], const float4 (&rp)[NP4],
                                          const int4 &rag, const float4 &rgs, const float4 &k1, const float4 &k2,
                                          const float4 &k3, const float4 &sc4, const float4 &sh4)
{
    constexpr int ZSTEP = 256 / (CO / 4), PSTEP = 256 / (CI / 4);  // staged by the 256 threads of the dgrad waves
    constexpr int LDZ = CO + 2, LDP = CI + 2;
    const int R = g.dz.rows;
    const int row0 = tile * TR;
    const int zc4 = (tid % (CO / 4)) * 4, zr = tid / (CO / 4);
    const int pc4 = (tid % (CI / 4)) * 4, pr = tid / (CI / 4);
#pragma unroll
    for (int q = 0; q < NZ4; ++q) {
        const int rt = zr + q * ZSTEP;
        float4 d;
        if (ZMODE == DZ_POOL) {
            const int n = n0 + rt;
            d.x = rag.x == n ? rgs.x : 0.f;
            d.y = rag.y == n ? rgs.y : 0.f;
            d.z = rag.z == n ? rgs.z : 0.f;
            d.w = rag.w == n ? rgs.w : 0.f;
        } else {
            d = rdy[q];
        }
        float4 v = make_float4(fmaf(k1.x, d.x, fmaf(k2.x, rz[q].x, k3.x)), fmaf(k1.y, d.y, fmaf(k2.y, rz[q].y, k3.y)),
                               fmaf(k1.z, d.z, fmaf(k2.z, rz[q].z, k3.z)), fmaf(k1.w, d.w, fmaf(k2.w, rz[q].w, k3.w)));
        if (!FULLR) {
            const float m = row0 + rt < R ? 1.f : 0.f;
            v.x *= m, v.y *= m, v.z *= m, v.w *= m;
        }
        float *o = Zs + rt * LDZ + zc4;
        *reinterpret_cast<float2 *>(o) = make_float2(v.x, v.y);
        *reinterpret_cast<float2 *>(o + 2) = make_float2(v.z, v.w);
    }
#pragma unroll
    for (int q = 0; q < NP4; ++q) {
        // the wgrad B operand is the ACTIVATION relu(bn(Zprev)): transformed here, once, by all eight waves (in the MFMA
        // loop the two VALU ops per fragment serialised with the wave's own MFMAs -- measured 3x slower)
        float *o = Ps + (pr + q * PSTEP) * LDP + pc4;
        *reinterpret_cast<float2 *>(o) = make_float2(relu_np(fmaf(rp[q].x, sc4.x, sh4.x)), relu_np(fmaf(rp[q].y, sc4.y, sh4.y)));
        *reinterpret_cast<float2 *>(o + 2) =
            make_float2(relu_np(fmaf(rp[q].z, sc4.z, sh4.z)), relu_np(fmaf(rp[q].w, sc4.w, sh4.w)));
    }
}

// Fragment fetch / MFMA groups of the fused kernel.  The MFMA loops are software-pipelined by hand: the LDS reads of
// group g+1 are issued before the MFMAs of group g, and a scheduling barrier after every group keeps the compiler from
// hoisting all reads to the top (which costs a live register per read) while still overlapping read latency with MFMAs.
template <int GS, bool WLDS, int LDW>
__device__ __forceinline__ void cbf_dg_load(float (&a)[GS], float (&b)[GS], const float *ap, const float *bp, int s0)
{
#pragma unroll
    for (int i = 0; i < GS; ++i) {
        a[i] = ap[2 * (s0 + i)];
        if (WLDS) b[i] = bp[2 * (s0 + i) * LDW];
    }
}

template <int GS, int NWT, int NCB, int LDZ, int LDP>
__device__ __forceinline__ void cbf_wg_load(float (&a)[GS], float (&b)[GS][NWT], const float *ap, const float *bp, int q0,
                                            int s0)
{
#pragma unroll
    for (int i = 0; i < GS; ++i) {
        a[i] = ap[2 * (s0 + i) * LDZ];
#pragma unroll
        for (int n = 0; n < NWT; ++n) b[i][n] = bp[2 * (s0 + i) * LDP + ((q0 + n) % NCB) * 32];
    }
}

template <int GS, int NWT>
__device__ __forceinline__ void cbf_wg_mfma(f32x16 (&acc)[NWT], const float (&a)[GS], const float (&b)[GS][NWT])
{
#pragma unroll
    for (int i = 0; i < GS; ++i)
#pragma unroll
        for (int n = 0; n < NWT; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[i][n], acc[n], 0, 0, 0);
}

// W[k = co][j = ci] -> LDS (pitch LDW), by all 512 threads
template <int CI, int CO, int LDW>
__device__ __forceinline__ void cbf_stage_w(const float *__restrict__ W, float *__restrict__ Ws, int tid)
{
    constexpr int W4 = CO * CI / 4 / 512;
    float4 wv[W4];
#pragma unroll
    for (int q = 0; q < W4; ++q) wv[q] = *reinterpret_cast<const float4 *>(W + (size_t)(tid + q * 512) * 4);
#pragma unroll
    for (int q = 0; q < W4; ++q) {
        const int f = tid + q * 512;
        *reinterpret_cast<float4 *>(Ws + (f / (CI / 4)) * LDW + (f % (CI / 4)) * 4) = wv[q];
    }
}

// Tile height and the home of W by shape: CO = 128 -> W in LDS (its 64 B-fragment registers per dgrad wave do not fit
// next to the prefetch registers); 128 x 128 channels -> 32-row tiles so that W and two tile buffers fit in 160 KB.
template <int CI, int CO>
struct CbfShape {
    static constexpr bool BOTH = CI == 128 && CO == 128;
    static constexpr int TR = BOTH ? 32 : 64;
    static constexpr bool WLDS = CO == 128;  // 64 B-fragment registers per dgrad wave otherwise
    static constexpr int LDW = CI + 4;
    static constexpr int LDZ = CO + 2, LDP = CI + 2;
    static constexpr int BUF = TR * (LDZ + LDP);
    static constexpr int WSZ = WLDS ? CO * LDW : 0;
    static constexpr int TSZ = 4 * 32 * 36;  // per dgrad wave: 32 x 32 output fragment, transposed for 16-byte stores
    static constexpr int XSZ = 2 * 3 * TR;   // IN3: the xyz rows of two tiles, coordinate-major [2][3][TR]
    static constexpr size_t LDS_BYTES = ((size_t)2 * BUF + WSZ + TSZ) * sizeof(float);
    static constexpr size_t LDS_BYTES_IN3 = LDS_BYTES + XSZ * sizeof(float);
};

// IN3: the layer below is the xyz input layer (3 input channels, conv_in3_fwd_kernel).  Its weight gradient
//   dW_in[c][d] = sum_r dZprev[r][c] x[r][d],   dZprev = k1 g + k2 Zprev + k3,  Zprev[r][c] = W_in[c] . x_r + b_in[c]
// needs no pass of its own over the 8 MB of g = dYprev: with Gx[c][d] = sum_r g[r][c] x[r][d] accumulated HERE (3 more
// sums per channel next to the two BatchNorm-backward sums) and the second moments of x,
//   dW_in[c][d] = k1 Gx[c][d] + k2 (sum_e W_in[c][e] Sxx[e][d] + b_in[c] Sx[d]) + k3 Sx[d]        (post_bwd_in3_kernel).
// Statistics partial per workgroup: [6][CI] = sum g, sum g Z, Gx[0..2], (Sx[3], Sxx[6] upper triangle, 0 ...).
template <int CI, int CO, int ZMODE, bool FULLR, bool IN3 = false>
__global__ void __launch_bounds__(512) conv_bwd_fused_kernel(ConvBwdArgs g)
{
    using S = CbfShape<CI, CO>;
    static_assert(!IN3 || (S::TR == 64 && ZMODE == DZ_BN), "IN3: 64-row tiles (one row per lane for the moments)");
    constexpr int NST = IN3 ? 5 : 2;  // per-channel sums of the dgrad epilogue
    constexpr int TR = S::TR, LDZ = S::LDZ, LDP = S::LDP, LDW = S::LDW;
    constexpr int ZB = TR * LDZ, BUF = S::BUF;
    constexpr int NZ4 = TR * CO / 4 / 256, NP4 = TR * CI / 4 / 256;  // float4 per dgrad-wave thread per tile
    constexpr bool WLDS = S::WLDS;
    constexpr int NCB = CI / 32, NOB = CO / 32, RB = TR / 32;
    constexpr int NDW = RB * NCB;          // dgrad tiles per row tile = dgrad waves (waves 0..3)
    constexpr int NWT = NOB * NCB / 4;     // wgrad tiles per wgrad wave (waves 4..7): one dW row block, NWT column blocks
    static_assert((CI == 64 || CI == 128) && (CO == 64 || CO == 128), "instantiated for 64 / 128 channels");
    static_assert(NDW == 4 && NZ4 >= 1 && NP4 >= 1 && NWT >= 1, "wave roles below assume four dgrad tiles per row tile");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *Ws = lds + 2 * BUF;  // [CO][LDW] when WLDS

    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float *Ts = lds + 2 * BUF + S::WSZ + (wave & 3) * (32 * 36);  // this dgrad wave's transpose scratch [32][36]
    float *Xs = lds + 2 * BUF + S::WSZ + S::TSZ;                    // IN3: [2][3][TR]
    const int R = g.dz.rows;
    const bool do_d = wave < 4;
    const int dwv = wave & 3;
    const int rb = dwv % RB, cb = dwv / RB;  // dgrad tile: rows rb*32.., channels cb*32..
    const int q0 = dwv * NWT;                // first wgrad tile of this wave
    const int cob = q0 / NCB;                // wgrad tiles: dW rows cob*32.., columns ((q0 + n) % NCB)*32..
    const int G = gridDim.x;

    SN_TL(0);
#ifdef SN_TIMELINE
    sn_hw_record();
#endif
    if (do_d) {
        // ---------------- producer + data-gradient waves ------------------------------------------------
        // They win the matrix-pipe arbitration (older waves), finish their 64-deep MFMA chain in about half a tile period
        // and spend the rest of it on the epilogue and on staging the NEXT tile into the other LDS buffer, while the
        // weight-gradient wave of the same SIMD still has the pipe busy.  (s_setprio for these waves: no effect, measured.)
        const int zc4 = (tid % (CO / 4)) * 4, pc4 = (tid % (CI / 4)) * 4;
        const bool fxin = ZMODE == DZ_BN && g.acc_in != nullptr;
        float4 k1 = make_float4(0.f, 0.f, 0.f, 0.f), k2 = k1, k3 = k1;
        if (!fxin) {
            k1 = *reinterpret_cast<const float4 *>(g.dz.k1 + zc4);
            k2 = *reinterpret_cast<const float4 *>(g.dz.k2 + zc4);
            k3 = *reinterpret_cast<const float4 *>(g.dz.k3 + zc4);
        }
        const float4 sc4 = *reinterpret_cast<const float4 *>(g.scale_prev + pc4);
        const float4 sh4 = *reinterpret_cast<const float4 *>(g.shift_prev + pc4);
        const float scd = g.scale_prev[cb * 32 + l31], shd = g.shift_prev[cb * 32 + l31];
        // byte offsets inside a tile that never change: fragment element 0 / transposed piece 0 of this lane, staging slots
        const unsigned qvo = ((rb * 32 + 4 * h) * CI + cb * 32 + l31) * 4;
        const unsigned ovo = ((rb * 32 + (lane >> 3)) * CI + cb * 32 + (lane & 7) * 4) * 4;
        const unsigned zvo = ((tid / (CO / 4)) * CO + zc4) * 4, pvo = ((tid / (CI / 4)) * CI + pc4) * 4, avo = zc4 * 4;
        CbfRsrc rs;
        rs.z = make_rsrc(g.dz.z, (unsigned)R * CO * 4);
        rs.dy = make_rsrc(ZMODE == DZ_BN ? g.dz.dy : g.dz.z, (unsigned)R * CO * 4);
        rs.zprev = make_rsrc(g.zprev, (unsigned)R * CI * 4);
        rs.dyprev = make_rsrc(g.dyprev, (unsigned)R * CI * 4);
        const unsigned nclouds = ZMODE == DZ_POOL ? (unsigned)((R + g.dz.npts - 1) / g.dz.npts) : 1u;
        rs.argsel = make_rsrc(ZMODE == DZ_POOL ? (const void *)g.dz.argsel : (const void *)g.dz.z, nclouds * CO * 4);
        rs.gsel = make_rsrc(ZMODE == DZ_POOL ? (const void *)g.dz.gsel : (const void *)g.dz.z, nclouds * CO * 4);
        // IN3: the tile's 3 TR input floats are one contiguous stretch: threads 0 .. 3 TR / 4 - 1 fetch 16 bytes each and
        // scatter them coordinate-major into LDS (fixed per-thread slots)
        const sn_rsrc rsx = make_rsrc(IN3 ? (const void *)g.xin : (const void *)g.dz.z, (unsigned)R * 12);
        const bool xthr = IN3 && tid < 3 * TR / 4;
        int xslot[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = tid * 4 + j;
            xslot[j] = (i % 3) * TR + i / 3;
        }
        float4 rx = make_float4(0.f, 0.f, 0.f, 0.f);
        float gx0 = 0.f, gx1 = 0.f, gx2 = 0.f;
        float mom[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) mom[k] = 0.f;
        // cloud b and tile-within-cloud of the current tile, advanced without divisions (DZ_POOL: one cloud per tile)
        const int tpc = ZMODE == DZ_POOL ? g.dz.npts / TR : 1;
        const int bstep = G / tpc, tstep = G - bstep * tpc;
        int cloud = (int)blockIdx.x / tpc, tic = (int)blockIdx.x - cloud * tpc;
        float4 rz[NZ4], rdy[NZ4], rp[NP4];
        int4 rag = make_int4(0, 0, 0, 0);
        float4 rgs = make_float4(0.f, 0.f, 0.f, 0.f);
        float s0 = 0.f, s1 = 0.f;
        // dYprev tile of the previous iteration, already transposed to 4 channels per lane, stored one iteration late:
        // lane L, piece i -> row 8 i + (L >> 3), channels 4 (L & 7) .. +3 of the wave's 32 x 32 block
        float4 vout[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) vout[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        constexpr int NWREG = WLDS ? 1 : CO / 2;
        float wreg[NWREG];

        int tile = blockIdx.x;
        cbf_issue_loads<CO, CI, TR, ZMODE, NZ4, NP4>(rs, tile, cloud, zvo, pvo, avo, rz, rdy, rp, rag, rgs);
        if (xthr) rx = buf_load4(rsx, (unsigned)tid * 16, (unsigned)tile * (TR * 12));
        if (WLDS) cbf_stage_w<CI, CO, LDW>(g.W, Ws, tid);  // requested after the first tile: its staging does not wait for W
        if (!WLDS) {  // dgrad B fragments in registers, k = 2 s + h (requested after the first tile)
#pragma unroll
            for (int s = 0; s < NWREG; ++s) wreg[s] = g.W[(size_t)(2 * s + h) * CI + cb * 32 + l31];
        }
        if (fxin) {
            // k1..k3 of this layer's BatchNorm backward: derived by the weight-gradient waves (idle until the first tile is
            // staged) from the fixed-point sums while the loads above are in flight
            const float *Ks = lds + 2 * BUF + S::WSZ;  // the transpose scratch is idle until the first epilogue
            __syncthreads();
            k1 = *reinterpret_cast<const float4 *>(Ks + zc4);
            k2 = *reinterpret_cast<const float4 *>(Ks + CO + zc4);
            k3 = *reinterpret_cast<const float4 *>(Ks + 2 * CO + zc4);
        }
        cbf_stage<CO, CI, TR, ZMODE, FULLR, NZ4, NP4>(g, tile, tic * TR, tid, lds, lds + ZB, rz, rdy, rp, rag, rgs, k1, k2, k3, sc4,
                                                      sh4);
        if (xthr) Xs[xslot[0]] = rx.x, Xs[xslot[1]] = rx.y, Xs[xslot[2]] = rx.z, Xs[xslot[3]] = rx.w;
        __syncthreads();
        for (int it = 0; tile < g.ntiles; ++it, tile += G) {
            const float *Zs = lds + (it & 1) * BUF;
            // dYprev of the previous tile goes out first: vmcnt retires in order, so stores issued after the loads below
            // would be waited for together with them
            if (!IN3 && it > 0) {  // (IN3: nobody reads dYprev -- the input layer's weight gradient comes from the sums below)
                const unsigned oso = (unsigned)(tile - G) * (TR * CI * 4);
#pragma unroll
                for (int i = 0; i < 4; ++i) buf_store4(vout[i], rs.dyprev, ovo + i * (8 * CI * 4), oso);
            }
            // raw Zprev at this wave's dYprev fragment positions (ReLU mask, BatchNorm-backward sum): from global memory
            // (L2-hot: the tile was fetched for the staging a moment ago), requested ahead of the MFMAs, used after them
            float zq[16];
#pragma unroll
            for (int e = 0; e < 16; ++e)
                zq[e] = buf_load1(rs.zprev, qvo + ((e & 3) + 8 * (e >> 2)) * (CI * 4), (unsigned)tile * (TR * CI * 4));
            // next tile's operands (the last iteration re-reads its own tile: keeps the loads unconditional)
            const bool more = tile + G < g.ntiles;
            const int nxt = more ? tile + G : tile;
            int ncloud = cloud, ntic = tic;
            if (more) {
                ncloud += bstep, ntic += tstep;
                if (ntic >= tpc) ntic -= tpc, ++ncloud;
            }
            cbf_issue_loads<CO, CI, TR, ZMODE, NZ4, NP4>(rs, nxt, ncloud, zvo, pvo, avo, rz, rdy, rp, rag, rgs);
            if (xthr) rx = buf_load4(rsx, (unsigned)tid * 16, (unsigned)nxt * (TR * 12));
            if (it == 1) SN_TL(5);

            f32x16 acc;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = 0.f;
            const float *ap = Zs + (rb * 32 + l31) * LDZ + h;
            const float *bp = Ws + h * LDW + cb * 32 + l31;
            constexpr int GD = 8, NGD = CO / 2 / GD;  // 8 k-steps per group
            static_assert(NGD % 2 == 0, "group count must be even");
            float a0[GD], b0[GD], a1[GD], b1[GD];
            cbf_dg_load<GD, WLDS, LDW>(a0, b0, ap, bp, 0);
#pragma unroll
            for (int gi = 0; gi < NGD; gi += 2) {
                cbf_dg_load<GD, WLDS, LDW>(a1, b1, ap, bp, (gi + 1) * GD);
                __builtin_amdgcn_sched_barrier(0);  // reads first, then the previous group's MFMAs
#pragma unroll
                for (int i = 0; i < GD; ++i)
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[i], WLDS ? b0[i] : wreg[WLDS ? 0 : gi * GD + i], acc, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (gi + 2 < NGD) cbf_dg_load<GD, WLDS, LDW>(a0, b0, ap, bp, (gi + 2) * GD);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < GD; ++i)
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[i], WLDS ? b1[i] : wreg[WLDS ? 0 : (gi + 1) * GD + i], acc, 0, 0,
                                                               0);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (it == 1) SN_TL(1);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float z = zq[e];
                const float v = fmaf(z, scd, shd) > 0.f ? acc[e] : 0.f;
                s0 += v;
                s1 += v * z;
                if (IN3) acc[e] = v;
                if (!IN3) Ts[frag_row(e, lane) * 36 + l31] = v;  // a dword store per fragment element costs ~58 issue cycles
            }                                                      // per wave-instruction: transpose in LDS, 16-byte stores
            if (IN3) {
                // rows of fragment elements 4 q .. 4 q + 3 are consecutive (frag_row): one 16-byte LDS read per coordinate
                const float *Xc = Xs + (it & 1) * (3 * TR);
                const float *xp = Xc + rb * 32 + 4 * h;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 x0 = *reinterpret_cast<const float4 *>(xp + 8 * q);
                    const float4 x1 = *reinterpret_cast<const float4 *>(xp + TR + 8 * q);
                    const float4 x2 = *reinterpret_cast<const float4 *>(xp + 2 * TR + 8 * q);
                    gx0 = fmaf(acc[4 * q + 3], x0.w, fmaf(acc[4 * q + 2], x0.z, fmaf(acc[4 * q + 1], x0.y, fmaf(acc[4 * q], x0.x, gx0))));
                    gx1 = fmaf(acc[4 * q + 3], x1.w, fmaf(acc[4 * q + 2], x1.z, fmaf(acc[4 * q + 1], x1.y, fmaf(acc[4 * q], x1.x, gx1))));
                    gx2 = fmaf(acc[4 * q + 3], x2.w, fmaf(acc[4 * q + 2], x2.z, fmaf(acc[4 * q + 1], x2.y, fmaf(acc[4 * q], x2.x, gx2))));
                }
                if (wave == 0) {  // moments of x: lane = row of the tile (rows past R were fetched as zeros)
                    const float a = Xc[lane], b = Xc[TR + lane], c = Xc[2 * TR + lane];
                    mom[0] += a, mom[1] += b, mom[2] += c;
                    mom[3] = fmaf(a, a, mom[3]), mom[4] = fmaf(a, b, mom[4]), mom[5] = fmaf(a, c, mom[5]);
                    mom[6] = fmaf(b, b, mom[6]), mom[7] = fmaf(b, c, mom[7]), mom[8] = fmaf(c, c, mom[8]);
                }
            }
            if (!IN3) {
#pragma unroll
                for (int i = 0; i < 4; ++i) vout[i] = *reinterpret_cast<const float4 *>(Ts + (8 * i + (lane >> 3)) * 36 + (lane & 7) * 4);
            }
            if (it == 1) SN_TL(2);
            if (more) {
                float *Zn = lds + ((it + 1) & 1) * BUF;
                cbf_stage<CO, CI, TR, ZMODE, FULLR, NZ4, NP4>(g, nxt, ntic * TR, tid, Zn, Zn + ZB, rz, rdy, rp, rag, rgs, k1, k2, k3,
                                                              sc4, sh4);
                if (xthr) {
                    float *Xn = Xs + ((it + 1) & 1) * (3 * TR);
                    Xn[xslot[0]] = rx.x, Xn[xslot[1]] = rx.y, Xn[xslot[2]] = rx.z, Xn[xslot[3]] = rx.w;
                }
            }
            cloud = ncloud, tic = ntic;
            if (it == 1) SN_TL(3);
            __syncthreads();
            if (it == 1) SN_TL(4);
        }
        SN_TL(6);
        if (!IN3 && tile != (int)blockIdx.x) {  // dYprev of the last tile
            const unsigned oso = (unsigned)(tile - G) * (TR * CI * 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) buf_store4(vout[i], rs.dyprev, ovo + i * (8 * CI * 4), oso);
        }
        // BatchNorm-backward sums of the layer below: halves of a wave, then the row blocks, fixed order
        float *red = lds;  // [RB][NST][CI]   (every wave is past its last LDS read: barrier at the end of the loop)
        const float t0 = s0 + __shfl_xor(s0, 32), t1 = s1 + __shfl_xor(s1, 32);
        if (lane < 32) {
            red[(rb * NST + 0) * CI + cb * 32 + lane] = t0;
            red[(rb * NST + 1) * CI + cb * 32 + lane] = t1;
        }
        if (IN3) {
            const float u0 = gx0 + __shfl_xor(gx0, 32), u1 = gx1 + __shfl_xor(gx1, 32), u2 = gx2 + __shfl_xor(gx2, 32);
            if (lane < 32) {
                red[(rb * NST + 2) * CI + cb * 32 + lane] = u0;
                red[(rb * NST + 3) * CI + cb * 32 + lane] = u1;
                red[(rb * NST + 4) * CI + cb * 32 + lane] = u2;
            }
            if (wave == 0) {
#pragma unroll
                for (int k = 0; k < 9; ++k) {
                    float m = mom[k];
#pragma unroll
                    for (int o = 32; o >= 1; o >>= 1) m += __shfl_xor(m, o);
                    if (lane == 0) red[RB * NST * CI + k] = m;
                }
            }
        }
    } else {
        // ---------------- weight-gradient waves ----------------------------------------------------------
        if (ZMODE == DZ_BN && g.acc_in != nullptr) {
            // this layer's BatchNorm backward from the fixed-point sums, for the dgrad waves' first staging
            float *Ks = lds + 2 * BUF + S::WSZ;
            const int c = tid - 256;
            if (c < CO) {
                const BnBwd bb = g.bb_in;
                double su, sz;
                fx_get2<kFxShiftBwd>(g.acc_in, c, su, sz);
                const BnBwdOut o = bn_backward_coefs(bb.R, su, sz, bn_bwd_inputs(bb, CO, c));
                Ks[c] = o.k1, Ks[CO + c] = o.k2, Ks[2 * CO + c] = o.k3;
                if (blockIdx.x == 0) {
                    bb.dgamma[c] = o.dgamma, bb.dbeta[c] = o.dbeta;
                    if (bb.dbias) bb.dbias[c] = o.dbias;
                    if (bb.kcoef) bb.kcoef[c] = o.k1, bb.kcoef[CO + c] = o.k2, bb.kcoef[2 * CO + c] = o.k3;
                }
            }
            fx_clear_share(g.zero_ptr, g.zero_n, blockIdx.x, gridDim.x, tid - 256, 256);
            __syncthreads();
        }
        if (WLDS) cbf_stage_w<CI, CO, LDW>(g.W, Ws, tid);
        f32x16 accw[NWT];
#pragma unroll
        for (int n = 0; n < NWT; ++n)
#pragma unroll
            for (int e = 0; e < 16; ++e) accw[n][e] = 0.f;
        __syncthreads();
        int tile = blockIdx.x;
        for (int it = 0; tile < g.ntiles; ++it, tile += G) {
            const float *Zs = lds + (it & 1) * BUF, *Ps = Zs + ZB;
            const float *ap = Zs + h * LDZ + cob * 32 + l31;
            const float *bp = Ps + h * LDP + l31;
            constexpr int GW = 2, NGW = TR / 2 / GW;  // 2 k-steps (NWT MFMAs each) per group
            static_assert(NGW % 2 == 0, "group count must be even");
            float a0[GW], b0[GW][NWT], a1[GW], b1[GW][NWT];
            cbf_wg_load<GW, NWT, NCB, LDZ, LDP>(a0, b0, ap, bp, q0, 0);
#pragma unroll
            for (int gi = 0; gi < NGW; gi += 2) {
                cbf_wg_load<GW, NWT, NCB, LDZ, LDP>(a1, b1, ap, bp, q0, (gi + 1) * GW);
                __builtin_amdgcn_sched_barrier(0);
                cbf_wg_mfma<GW, NWT>(accw, a0, b0);
                __builtin_amdgcn_sched_barrier(0);
                if (gi + 2 < NGW) cbf_wg_load<GW, NWT, NCB, LDZ, LDP>(a0, b0, ap, bp, q0, (gi + 2) * GW);
                __builtin_amdgcn_sched_barrier(0);
                cbf_wg_mfma<GW, NWT>(accw, a1, b1);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (it == 1) SN_TL(1);
            __syncthreads();
            if (it == 1) SN_TL(4);
        }
        SN_TL(6);
        // weight-gradient partial of this workgroup: each 32 x 32 fragment transposed through LDS (the tile buffers are
        // dead: every wave is past the loop's last barrier), 4 x 16-byte stores per lane instead of 16 dword stores
        float *P = g.part + (size_t)blockIdx.x * CO * CI;
        float *Tw = lds + RB * NST * CI + 16 + (wave - 4) * (32 * 36);  // behind the dgrad waves' statistics area
#pragma unroll
        for (int n = 0; n < NWT; ++n) {
            const int colb = ((q0 + n) % NCB) * 32;
#pragma unroll
            for (int e = 0; e < 16; ++e) Tw[frag_row(e, lane) * 36 + l31] = accw[n][e];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int rt = 8 * i + (lane >> 3);
                *reinterpret_cast<float4 *>(P + (size_t)(cob * 32 + rt) * CI + colb + (lane & 7) * 4) =
                    *reinterpret_cast<const float4 *>(Tw + rt * 36 + (lane & 7) * 4);
            }
        }
    }
    __syncthreads();
    if (tid < CI) {
        const float *red = lds;
        float *st = g.stats + (size_t)blockIdx.x * (IN3 ? 6 : 2) * CI;
#pragma unroll
        for (int k = 0; k < NST; ++k) {
            float a = red[k * CI + tid];
            if (RB == 2) a += red[(NST + k) * CI + tid];
            if (!IN3 && g.acc_out)
                fx_add<kFxShiftBwd>(g.acc_out, blockIdx.x % kFxSlots, k, tid, a);
            else
                st[k * CI + tid] = a;
        }
        if (IN3) st[5 * CI + tid] = tid < 9 ? red[RB * NST * CI + tid] : 0.f;
    }
    SN_TL_DRAIN();
    SN_TL(7);
}

// ------------------------------------------------------------------------------------------------
// conv_bwd_fused_kernel on the bf16 matrix cores (split-bf16 products, see gemm_tile_bx3): same walk over the row tiles, same
// wave roles, same epilogues and outputs.  What changes:
//  * the tile buffers hold three bf16 planes of dZ [TR][CO] and of the activation relu(bn(Zprev)) [TR][CI], row-major
//    (pitch + 8 elements): the producer waves split every element once, on its way into LDS;
//  * dgrad (K = co): A fragment = 16-byte reads of a dZ row; B = W^T, split and kept in registers for the whole kernel
//    (3 * CO / 16 fragments of 8 bf16 per lane);
//  * wgrad (K = tile rows): both operands are needed k(row)-major -- the transposing LDS read ds_read_b64_tr_b16 delivers,
//    from the same row-major images, 4 consecutive rows of one channel per lane (within a 16-lane group, lane l supplies
//    the address of row (l >> 2), channels 4 (l & 3) .. +3 and receives channel l, rows 0..3: checked on the hardware);
//  * six MFMAs (32 cycles each) per K = 16 instead of eight fp32 ones (64 cycles each).
// 64 -> 128 channels: 32-row tiles (two buffers of three planes must fit 160 KB), hence only two 32 x 32 dgrad tiles per row tile:
// the four dgrad waves pair up on a tile, each takes half of K, and the upper half's partial tile is added through LDS.
// ------------------------------------------------------------------------------------------------
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
__device__ __forceinline__ bf16x8 lds_tr8(const __bf16 *p, int pitch)  // rows r .. r+3 and r+4 .. r+7 of this lane's channel
{
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(p));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(p + 4 * pitch));
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}

__device__ __forceinline__ bf16x8 lds_tr8_2(const __bf16 *plo, const __bf16 *phi)  // the two halves from separate addresses
{
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(plo));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(phi));
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}

#ifndef SN_CBX_PW
#define SN_CBX_PW 3  // bit 0: the plain layers, bit 1: the layer above the xyz layer (the activation tile rebuilt from the coordinates)
#endif
#ifndef SN_CBX_PD
#define SN_CBX_PD 1  // (2: tile t + 2 of the activation requested while t + 1 waits -- measured SLOWER, 909 -> 1050 us at B = 2048)
#endif
#ifndef SN_CBX_DG2
#define SN_CBX_DG2 0  // (1: two accumulator chains in the data gradient -- measured equal: 939 vs 945 us at B = 2048, 21.3 vs 21.5 at B = 32)
#endif
#ifndef SN_CBX_ABL
#define SN_CBX_ABL 0  // (timing experiments only: 1 no data-gradient MFMAs, 2 no weight-gradient MFMAs / fragment reads, 3 no dYprev stores,
#endif                //  4 no staging, 5 no data-gradient epilogue, 6 no Zprev re-loads for the epilogue -- results are then garbage)
#ifndef SN_CBX_SWZ
#define SN_CBX_SWZ 1  // conflict-free LDS layout of the 128 x 128 kernel's tile planes (CbxShape::SWZ); 0: the round-5 layout (A/B)
#endif
#ifndef SN_CBX_SKEW
#define SN_CBX_SKEW 1  // the 128 x 128 kernel's two wave groups half a tile apart (see kSkew in conv_bwd_bx3_kernel); 0: in lockstep (A/B)
#endif
#ifndef SN_CBX_LATE_ST
#define SN_CBX_LATE_ST 0  // (1: the previous tile's dYprev stores behind the first k-step's MFMAs, see store_prev -- measured: 128 x 128 772 -> 765 us,
                          //  64 -> 128 616 -> 648, 64 x 64 376 -> 395 at B = 2048: the acknowledgements are not what the waves wait for)
#endif
#ifndef SN_CBX_SKEW_DG2
#define SN_CBX_SKEW_DG2 0  // (1: two accumulator chains in the skewed kernel's data gradient -- measured equal, 795 vs 784-794 us: one chain
                          //  issues at the pipe's rate, tools/micro/mfma_bf16_chain.hip; 0 keeps the round-5 kernel's bits)
#endif
#ifndef SN_CBX_CONTIG
#define SN_CBX_CONTIG 0  // (1: a contiguous range of tiles per workgroup -- measured equal to -6 %)
#endif
template <int CI, int CO>
struct CbxShape {
    static constexpr int TR = CO == 128 ? 32 : 64;                  // two tile buffers of three planes must fit the LDS
    static constexpr int KS = (CI == 64 && CO == 128) ? 2 : 1;      // 32 x 64 dgrad block = two 32 x 32 tiles: four waves split K too
    // bf16 pitches.  SWZ (128 x 128 tiles): 320-byte rows -- the four consecutive rows a 32-lane group of a transposing read touches
    // land on four disjoint 16-bank windows (272-byte rows: 4-way conflicts, half of the kernel's LDS cycles) -- and, in the dZ planes,
    // the 16-byte slots of a row XOR-ed with (row / 4) % 4, which keeps the data-gradient waves' row-wise 16-byte reads (lane groups
    // {0-3, 12-15, 20-27}, ...) conflict-free at that pitch; the XOR stays inside a 64-byte window, so the transposing reads keep theirs
    static constexpr bool SWZ = SN_CBX_SWZ != 0 && CO == 128;
    static constexpr int LDZ = SWZ ? 160 : CO + 8, LDP = SWZ ? (CI == 128 ? 160 : 96) : CI + 8;  // (96: 192-byte rows, windows 0, 48, 32, 16)
    static constexpr int ZPL = TR * LDZ, PPL = TR * LDP;            // one plane
    static constexpr int BUF = 3 * (ZPL + PPL);                     // bf16 elements per tile buffer
    static constexpr int TSZ = 4 * 32 * 36;                         // floats: per dgrad wave 32 x 32 transpose scratch
    static constexpr int XSZ = 2 * 3 * TR;
    static constexpr size_t TOFF = (size_t)2 * BUF * 2;             // byte offset of the float areas behind the tile buffers
    static constexpr size_t LDS_BYTES = TOFF + TSZ * sizeof(float);
    static constexpr size_t LDS_BYTES_IN3 = LDS_BYTES + (XSZ + 4 * CI) * sizeof(float);  // + coordinate rows, xyz-layer parameters (RZ1)
    static_assert(LDS_BYTES_IN3 <= 160 * 1024, "tile buffers exceed the LDS");
};

// RZ1: rp[q] holds the xyz coordinates of the row; w3 -> the xyz layer's (w0, w1, w2, bias) of this thread's four channels, in LDS
template <int CO, int CI, int TR, int ZMODE, bool FULLR, int NZ4, int NP4, bool RZ1 = false, bool SKIP_PS = false>
__device__ __forceinline__ void cbx_stage(const ConvBwdArgs &g, int tile, int n0, int tid, __bf16 *__restrict__ Zb,
                                          __bf16 *__restrict__ Pb, const float4 (&rz)[NZ4], const float4 (&rdy)[NZ4],
                                          const float4 (&rp)[NP4], const int4 &rag, const float4 &rgs, const float4 &k1,
                                          const float4 &k2, const float4 &k3, const float4 &sc4, const float4 &sh4,
                                          const float4 *w3 = nullptr)
{
    constexpr int ZSTEP = 256 / (CO / 4), PSTEP = 256 / (CI / 4);
    constexpr int LDZ = CbxShape<CI, CO>::LDZ, LDP = CbxShape<CI, CO>::LDP;
    const int R = g.dz.rows;
    const int row0 = tile * TR;
    const int zc4 = (tid % (CO / 4)) * 4, zr = tid / (CO / 4);
    const int pc4 = (tid % (CI / 4)) * 4, pr = tid / (CI / 4);
#pragma unroll
    for (int q = 0; q < NZ4; ++q) {
        const int rt = zr + q * ZSTEP;
        float4 d;
        if (ZMODE == DZ_POOL) {
            const int n = n0 + rt;
            d.x = rag.x == n ? rgs.x : 0.f;
            d.y = rag.y == n ? rgs.y : 0.f;
            d.z = rag.z == n ? rgs.z : 0.f;
            d.w = rag.w == n ? rgs.w : 0.f;
        } else {
            d = rdy[q];
        }
        float4 v = make_float4(fmaf(k1.x, d.x, fmaf(k2.x, rz[q].x, k3.x)), fmaf(k1.y, d.y, fmaf(k2.y, rz[q].y, k3.y)),
                               fmaf(k1.z, d.z, fmaf(k2.z, rz[q].z, k3.z)), fmaf(k1.w, d.w, fmaf(k2.w, rz[q].w, k3.w)));
        if (!FULLR) {
            const float m = row0 + rt < R ? 1.f : 0.f;
            v.x *= m, v.y *= m, v.z *= m, v.w *= m;
        }
        stage_split_p<TR * LDZ, LDZ>(Zb, rt, CbxShape<CI, CO>::SWZ ? zc4 ^ (((rt >> 2) & 3) << 3) : zc4, v);
    }
#pragma unroll
    for (int q = 0; q < (SKIP_PS ? 0 : NP4); ++q) {
        float4 zp = rp[q];
        if (RZ1) {
#pragma clang fp contract(off)
            const float x0 = rp[q].x, x1 = rp[q].y, x2 = rp[q].z;
            const float4 c0 = w3[0], c1 = w3[1], c2 = w3[2], c3 = w3[3];
            zp.x = fmaf(c0.z, x2, fmaf(c0.y, x1, c0.x * x0)) + c0.w;
            zp.y = fmaf(c1.z, x2, fmaf(c1.y, x1, c1.x * x0)) + c1.w;
            zp.z = fmaf(c2.z, x2, fmaf(c2.y, x1, c2.x * x0)) + c2.w;
            zp.w = fmaf(c3.z, x2, fmaf(c3.y, x1, c3.x * x0)) + c3.w;
        }
        const float4 a = make_float4(relu_np(fmaf(zp.x, sc4.x, sh4.x)), relu_np(fmaf(zp.y, sc4.y, sh4.y)),
                                     relu_np(fmaf(zp.z, sc4.z, sh4.z)), relu_np(fmaf(zp.w, sc4.w, sh4.w)));
        stage_split_p<TR * LDP, LDP>(Pb, pr + q * PSTEP, pc4, a);
    }
}

// RZ1 (IN3 only): Zprev is not read -- the producer rebuilds it from the tile's xyz rows (ConvBwdArgs::w_in)
// GZ / GP / GW: global row strides (elements) of the dZ-side tensors (Z, dY), of Zprev / dYprev and of W -- a layer with 256 channels on one
// side runs as two passes of the 128 x 128 instantiation over the halves of that side (the reconstruction sampler's 128 -> 256 -> 128):
//   256 output channels: the passes take dZ columns / W rows [0,128) and [128,256); the data gradient is their SUM -- DM = 1 (first pass)
//     stores it raw (no ReLU mask, no statistics), DM = 2 (second) adds ConvBwdArgs::dyacc at the fragment positions before the epilogue;
//   256 input channels: the passes take W / Zprev / dYprev columns [0,128) and [128,256) and are independent (DM = 0).
// PW: the activation tile relu(bn(Zprev)) is fetched, activated, split and staged by the WEIGHT-GRADIENT waves (behind their
//   MFMAs, where they used to wait ~1.5 us per tile at the barrier for the producer waves: tools/timeline.py bwd), the dZ tile
//   stays with the data-gradient waves -- the serial chain MFMAs -> epilogue -> staging of those waves loses its Zprev half
template <int CI, int CO, int ZMODE, bool FULLR, bool IN3 = false, bool RZ1 = false, int GZ = CO, int GP = CI, int GW = CI, int DM = 0,
          bool PW = (SN_CBX_PW != 0) && (!IN3 || (SN_CBX_PW & 2) != 0)>
__global__ void __launch_bounds__(512) conv_bwd_bx3_kernel(ConvBwdArgs g)
{
    static_assert(DM == 0 || (!IN3 && CbxShape<CI, CO>::KS == 1), "two-pass modes: plain 128 x 128 tiles");
    static_assert(!RZ1 || IN3, "RZ1: the layer below must be the xyz layer");
    using S = CbxShape<CI, CO>;
    static_assert(!IN3 || (S::TR == 64 && ZMODE == DZ_BN), "IN3: 64-row tiles (one row per lane for the moments)");
    constexpr int NST = IN3 ? 5 : 2;
    constexpr int TR = S::TR, LDZ = S::LDZ, LDP = S::LDP, ZPL = S::ZPL, PPL = S::PPL, BUF = S::BUF;
    constexpr int NZ4 = TR * CO / 4 / 256, NP4 = TR * CI / 4 / 256;
    constexpr int NCB = CI / 32, NOB = CO / 32, RB = TR / 32;
    constexpr int KS = S::KS;              // dgrad waves per 32 x 32 tile (each takes a K range; summed through LDS)
    // kSkew (128 x 128 tiles): the two wave groups run half a tile apart -- slot A: the data-gradient waves' MFMAs (they own the
    // matrix pipe in this slot) beside the weight-gradient waves' staging of the next activation tile; slot B: the
    // data-gradient epilogue + staging of the next dZ tile beside the weight-gradient MFMAs; one barrier per slot.  In lockstep (one
    // barrier per tile) both groups' MFMAs share the pipe for 1.4 us and then both groups run VALU work with the pipe idle.
    constexpr bool kSkew = SN_CBX_SKEW != 0 && PW && !IN3 && CI == 128 && CO == 128;
    constexpr bool kDg2 = SN_CBX_DG2 != 0 || (kSkew && SN_CBX_SKEW_DG2 != 0);
    constexpr int NDW = RB * NCB * KS;
    constexpr int NWT = NOB * NCB / 4;
    constexpr int KD = CO / 16 / KS, KW = TR / 16;  // K = 16 steps of a dgrad wave / of a row tile's wgrad
    static_assert((CI == 64 || CI == 128) && (CO == 64 || CO == 128), "instantiated for 64 / 128 channels");
    static_assert(NDW == 4 && NZ4 >= 1 && NP4 >= 1 && NWT >= 1, "wave roles below assume four dgrad waves per row tile");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __bf16 *Lb = reinterpret_cast<__bf16 *>(lds);
    float *Tf = reinterpret_cast<float *>(reinterpret_cast<char *>(lds) + S::TOFF);  // float areas behind the tile buffers

    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float *Ts = Tf + (wave & 3) * (32 * 36);
    float *Xs = Tf + S::TSZ;
    const int R = g.dz.rows;
    const bool do_d = wave < 4;
    const int dwv = wave & 3;
    const int dt = dwv / KS, kh = dwv % KS;  // dgrad tile of this wave and its K range
    const int rb = dt % RB, cb = dt / RB;
    const int q0 = dwv * NWT;
    const int cob = q0 / NCB;
    const int G = gridDim.x;
    // the tiles of a workgroup: a contiguous range (SN_CBX_CONTIG: sequential addresses per workgroup) or every G-th one
    constexpr bool CT = SN_CBX_CONTIG != 0;
    const int tpw = (g.ntiles + G - 1) / G;
    const int tile0 = CT ? (int)blockIdx.x * tpw : (int)blockIdx.x;
    const int tend = CT ? min(g.ntiles, tile0 + tpw) : g.ntiles;
    const int tst = CT ? 1 : G;

    SN_TL(0);
#ifdef SN_TIMELINE
    sn_hw_record();
#endif
    if (do_d) {
        // ---------------- producer + data-gradient waves ------------------------------------------------
        const int zc4 = (tid % (CO / 4)) * 4, pc4 = (tid % (CI / 4)) * 4;
        const bool fxin = ZMODE == DZ_BN && g.acc_in != nullptr;
        float4 k1 = make_float4(0.f, 0.f, 0.f, 0.f), k2 = k1, k3 = k1;
        if (!fxin) {
            k1 = *reinterpret_cast<const float4 *>(g.dz.k1 + zc4);
            k2 = *reinterpret_cast<const float4 *>(g.dz.k2 + zc4);
            k3 = *reinterpret_cast<const float4 *>(g.dz.k3 + zc4);
        }
        const float4 sc4 = *reinterpret_cast<const float4 *>(g.scale_prev + pc4);
        const float4 sh4 = *reinterpret_cast<const float4 *>(g.shift_prev + pc4);
        const float scd = g.scale_prev[cb * 32 + l31], shd = g.shift_prev[cb * 32 + l31];
        const unsigned qvo = ((rb * 32 + 4 * h) * GP + cb * 32 + l31) * 4;
        const unsigned ovo = ((rb * 32 + (lane >> 3)) * GP + cb * 32 + (lane & 7) * 4) * 4;
        const unsigned zvo = ((tid / (CO / 4)) * GZ + zc4) * 4, pvo = ((tid / (CI / 4)) * GP + pc4) * 4, avo = zc4 * 4;
        CbfRsrc rs;
        // (a pass over one half of a 256-channel side starts GZ / 2 or GP / 2 elements into the first row: the last row's range ends
        //  that far behind the tensor -- never touched, every lane stays inside its half)
        rs.z = make_rsrc(g.dz.z, (unsigned)R * GZ * 4);
        rs.dy = make_rsrc(ZMODE == DZ_BN ? g.dz.dy : g.dz.z, (unsigned)R * GZ * 4);
        rs.zprev = make_rsrc(g.zprev, (unsigned)R * GP * 4);
        rs.dyprev = make_rsrc(g.dyprev, (unsigned)R * GP * 4);
        rs.dyacc = make_rsrc(DM == 2 ? (const void *)g.dyacc : (const void *)g.zprev, (unsigned)R * GP * 4);
        const unsigned nclouds = ZMODE == DZ_POOL ? (unsigned)((R + g.dz.npts - 1) / g.dz.npts) : 1u;
        rs.argsel = make_rsrc(ZMODE == DZ_POOL ? (const void *)g.dz.argsel : (const void *)g.dz.z, nclouds * CO * 4);
        rs.gsel = make_rsrc(ZMODE == DZ_POOL ? (const void *)g.dz.gsel : (const void *)g.dz.z, nclouds * CO * 4);
        const sn_rsrc rsx = make_rsrc(IN3 ? (const void *)g.xin : (const void *)g.dz.z, (unsigned)R * 12);
        const bool xthr = IN3 && tid < 3 * TR / 4;
        int xslot[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = tid * 4 + j;
            xslot[j] = (i % 3) * TR + i / 3;
        }
        float4 rx = make_float4(0.f, 0.f, 0.f, 0.f);
        float gx0 = 0.f, gx1 = 0.f, gx2 = 0.f;
        float mom[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) mom[k] = 0.f;
        const int tpc = ZMODE == DZ_POOL ? g.dz.npts / TR : 1;
        const int bstep = tst / tpc, tstep = tst - bstep * tpc;
        int cloud = tile0 / tpc, tic = tile0 - cloud * tpc;
        float4 rz[NZ4], rdy[NZ4], rp[NP4];
        int4 rag = make_int4(0, 0, 0, 0);
        float4 rgs = make_float4(0.f, 0.f, 0.f, 0.f);
        float s0 = 0.f, s1 = 0.f;
        // RZ1: the xyz layer's weights of the four channels this thread stages and of the channel its dYprev fragment column holds
        // (w0, w1, w2, bias) per channel of the xyz layer, in LDS behind the coordinate rows: read at every use (20 registers otherwise)
        float4 *W3s = reinterpret_cast<float4 *>(Xs + S::XSZ);
        constexpr int PSTEPK = 256 / (CI / 4);
        const unsigned xvo = (tid / (CI / 4)) * 12;
        if (RZ1 && tid < CI)
            W3s[tid] = make_float4(g.w_in[tid * 3], g.w_in[tid * 3 + 1], g.w_in[tid * 3 + 2], g.b_in ? g.b_in[tid] : 0.f);
        const float4 *w3s = W3s + pc4;
        // (RZ1) the rows' coordinates in place of the Zprev tile: 12 bytes per row instead of 16 per four channels
        auto load_xyz_rows = [&](int t) __attribute__((always_inline)) {
#pragma unroll
            for (int q = 0; q < NP4; ++q) rp[q] = buf_load3(rsx, xvo, (unsigned)t * (TR * 12) + q * (PSTEPK * 12));
        };
        float4 vout[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) vout[i] = make_float4(0.f, 0.f, 0.f, 0.f);

        int tile = tile0;
        cbf_issue_loads<CO, CI, TR, ZMODE, NZ4, NP4, RZ1 || PW, GZ, GP>(rs, tile, cloud, zvo, pvo, avo, rz, rdy, rp, rag, rgs);
        if (RZ1 && !PW) load_xyz_rows(tile);
        if (xthr) rx = buf_load4(rsx, (unsigned)tid * 16, (unsigned)tile * (TR * 12));
        // W^T fragments of this wave's 32 input channels: W[co = 16 kk + 8 h + t][ci = cb 32 + l31] (requested after the first
        // tile: its staging does not wait for them), split below once the first tile is staged
        float wraw[KD][8];
#pragma unroll
        for (int kk = 0; kk < KD; ++kk)
#pragma unroll
            for (int t = 0; t < 8; ++t) wraw[kk][t] = g.W[(size_t)((kh * KD + kk) * 16 + 8 * h + t) * GW + cb * 32 + l31];
        if (fxin) {
            const float *Ks = Tf;  // the transpose scratch is idle until the first epilogue
            __syncthreads();
            k1 = *reinterpret_cast<const float4 *>(Ks + zc4);
            k2 = *reinterpret_cast<const float4 *>(Ks + CO + zc4);
            k3 = *reinterpret_cast<const float4 *>(Ks + 2 * CO + zc4);
        }
        cbx_stage<CO, CI, TR, ZMODE, FULLR, NZ4, NP4, RZ1, PW>(g, tile, tic * TR, tid, Lb, Lb + 3 * ZPL, rz, rdy, rp, rag, rgs, k1, k2, k3, sc4,
                                                               sh4, w3s);
        if (xthr) Xs[xslot[0]] = rx.x, Xs[xslot[1]] = rx.y, Xs[xslot[2]] = rx.z, Xs[xslot[3]] = rx.w;
        bf16x8 wf[KD][3];
#pragma unroll
        for (int kk = 0; kk < KD; ++kk)
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                __bf16 h1, h2, h3;
                split3(wraw[kk][t], h1, h2, h3);
                wf[kk][0][t] = h1, wf[kk][1][t] = h2, wf[kk][2][t] = h3;
            }
        __syncthreads();
        for (int it = 0; tile < tend; ++it, tile += tst) {
            const __bf16 *Zb = Lb + (it & 1) * BUF;
            // the previous tile's dYprev leaves at the top of the iteration.  (The register allocator gives the accumulator the registers
            // of `vout`, dead once stored, so the first MFMA waits for the four stores' acknowledgements -- s_waitcnt vmcnt on a store-data
            // hazard.  SN_CBX_LATE_ST=1 issues them behind the first k-step's MFMAs instead, from registers of their own: measured, and
            // not faster -- the write-back L2 acknowledges at once.)
            const auto store_prev = [&]() __attribute__((always_inline)) {
                if (!IN3 && it > 0 && kh == 0 && SN_CBX_ABL != 3) {
                    const unsigned oso = (unsigned)(tile - tst) * (TR * GP * 4);
#pragma unroll
                    for (int i = 0; i < 4; ++i) buf_store4(vout[i], rs.dyprev, ovo, oso + i * (8 * GP * 4));
                }
            };
            if (!SN_CBX_LATE_ST || SN_CBX_ABL == 1) store_prev();
            float zq[16], pq[16];
            if (!RZ1 && DM != 1 && (KS == 1 || kh == 0))
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    zq[e] = SN_CBX_ABL == 6 ? 1.0f : buf_load1(rs.zprev, qvo, (unsigned)tile * (TR * GP * 4) + ((e & 3) + 8 * (e >> 2)) * (GP * 4));
            if (DM == 2)  // the first pass's raw data gradient at this lane's fragment positions
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    pq[e] = buf_load1(rs.dyacc, qvo, (unsigned)tile * (TR * GP * 4) + ((e & 3) + 8 * (e >> 2)) * (GP * 4));
            const bool more = tile + tst < tend;
            const int nxt = more ? tile + tst : tile;
            int ncloud = cloud, ntic = tic;
            if (more) {
                ncloud += bstep, ntic += tstep;
                if (ntic >= tpc) ntic -= tpc, ++ncloud;
            }
            cbf_issue_loads<CO, CI, TR, ZMODE, NZ4, NP4, RZ1 || PW, GZ, GP>(rs, nxt, ncloud, zvo, pvo, avo, rz, rdy, rp, rag, rgs);
            if (RZ1 && !PW) load_xyz_rows(nxt);
            if (xthr) rx = buf_load4(rsx, (unsigned)tid * 16, (unsigned)nxt * (TR * 12));
            // the requests go out HERE: left alone, the scheduler sinks them below the MFMAs to their first use (the staging),
            // and every tile pays a full memory round trip
            __builtin_amdgcn_sched_barrier(0);
            if (it == 1) SN_TL(5);

            // two accumulator chains (even / odd k-steps, added at the end): one chain of KD x 6 dependent MFMAs issues at the
            // accumulator's latency (~70 cycles apiece, 1.4 us per 128-channel tile: tools/timeline.py bwd), not at the pipe's rate
            f32x16 acc, acc2;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = 0.f, acc2[e] = 0.f;
            // (SWZ: slot 2 kk + h of the row sits at slot (2 kk + h) ^ f, f = (row / 4) % 4: two base pointers, even and odd k-steps)
            const int fsw = S::SWZ ? ((l31 >> 2) & 3) ^ h : 0;
            const __bf16 *apr = Zb + (rb * 32 + l31) * LDZ + kh * KD * 16;
            const __bf16 *ape = S::SWZ ? apr + (fsw << 3) : apr + 8 * h, *apo = S::SWZ ? apr + ((fsw ^ 2) << 3) : apr + 8 * h + 16;
#pragma unroll
            for (int kk = 0; kk < (SN_CBX_ABL == 1 ? 0 : KD); ++kk) {
                const __bf16 *ap = ((kk & 1) ? apo : ape) + (kk >> 1) * 32;
                const bf16x8 a0 = *reinterpret_cast<const bf16x8 *>(ap);
                const bf16x8 a1 = *reinterpret_cast<const bf16x8 *>(ap + ZPL);
                const bf16x8 a2 = *reinterpret_cast<const bf16x8 *>(ap + 2 * ZPL);
                if (kDg2 && (kk & 1)) {
                    acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, wf[kk][2], acc2, 0, 0, 0);
                    acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, wf[kk][0], acc2, 0, 0, 0);
                    acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, wf[kk][1], acc2, 0, 0, 0);
                    acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, wf[kk][1], acc2, 0, 0, 0);
                    acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, wf[kk][0], acc2, 0, 0, 0);
                    acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, wf[kk][0], acc2, 0, 0, 0);
                } else {
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, wf[kk][2], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, wf[kk][0], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, wf[kk][1], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, wf[kk][1], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, wf[kk][0], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, wf[kk][0], acc, 0, 0, 0);
                }
                if (SN_CBX_LATE_ST && kk == 0) {
                    __builtin_amdgcn_sched_barrier(0);
                    store_prev();
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (kDg2)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[e] += acc2[e];
            if (it == 1) SN_TL(1);
            if (kSkew) __syncthreads();  // end of slot A
            if (KS == 2) {  // the upper K range's partial tile joins the lower one's through the upper wave's scratch
                float *Tx = Tf + (dwv | 1) * (32 * 36);
                if (kh == 1)
#pragma unroll
                    for (int e = 0; e < 16; ++e) Tx[e * 64 + lane] = acc[e];
                __syncthreads();
                if (kh == 0)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[e] += Tx[e * 64 + lane];
            }
            if (kh == 0 && SN_CBX_ABL != 5) {
            if (RZ1) {  // Zprev at the fragment positions from the tile's coordinates in LDS (rows 4 q .. 4 q + 3 of a fragment are consecutive)
#pragma clang fp contract(off)  // bit for bit the stored tensor: the bias add must not fuse with what consumes z below
                const float4 wd = W3s[cb * 32 + l31];
                const float w3d0 = wd.x, w3d1 = wd.y, w3d2 = wd.z, b3d = wd.w;
                const float *xq = Xs + (it & 1) * (3 * TR) + rb * 32 + 4 * h;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 x0 = *reinterpret_cast<const float4 *>(xq + 8 * q);
                    const float4 x1 = *reinterpret_cast<const float4 *>(xq + TR + 8 * q);
                    const float4 x2 = *reinterpret_cast<const float4 *>(xq + 2 * TR + 8 * q);
                    zq[4 * q + 0] = fmaf(w3d2, x2.x, fmaf(w3d1, x1.x, w3d0 * x0.x)) + b3d;
                    zq[4 * q + 1] = fmaf(w3d2, x2.y, fmaf(w3d1, x1.y, w3d0 * x0.y)) + b3d;
                    zq[4 * q + 2] = fmaf(w3d2, x2.z, fmaf(w3d1, x1.z, w3d0 * x0.z)) + b3d;
                    zq[4 * q + 3] = fmaf(w3d2, x2.w, fmaf(w3d1, x1.w, w3d0 * x0.w)) + b3d;
                }
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                if (DM == 1) {  // first of two passes over the output channels: the raw partial sum
                    Ts[frag_row(e, lane) * 36 + l31] = acc[e];
                    continue;
                }
                const float z = zq[e];
                const float a = DM == 2 ? acc[e] + pq[e] : acc[e];
                const float v = fmaf(z, scd, shd) > 0.f ? a : 0.f;
                s0 += v;
                s1 = fmaf(v, z, s1);  // (explicit: the variants of this kernel must round the sum the same way)
                if (IN3) acc[e] = v;
                if (!IN3) Ts[frag_row(e, lane) * 36 + l31] = v;
            }
            if (IN3) {
                const float *Xc = Xs + (it & 1) * (3 * TR);
                const float *xp = Xc + rb * 32 + 4 * h;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 x0 = *reinterpret_cast<const float4 *>(xp + 8 * q);
                    const float4 x1 = *reinterpret_cast<const float4 *>(xp + TR + 8 * q);
                    const float4 x2 = *reinterpret_cast<const float4 *>(xp + 2 * TR + 8 * q);
                    gx0 = fmaf(acc[4 * q + 3], x0.w, fmaf(acc[4 * q + 2], x0.z, fmaf(acc[4 * q + 1], x0.y, fmaf(acc[4 * q], x0.x, gx0))));
                    gx1 = fmaf(acc[4 * q + 3], x1.w, fmaf(acc[4 * q + 2], x1.z, fmaf(acc[4 * q + 1], x1.y, fmaf(acc[4 * q], x1.x, gx1))));
                    gx2 = fmaf(acc[4 * q + 3], x2.w, fmaf(acc[4 * q + 2], x2.z, fmaf(acc[4 * q + 1], x2.y, fmaf(acc[4 * q], x2.x, gx2))));
                }
                if (wave == 0) {
                    const float a = Xc[lane], b = Xc[TR + lane], c = Xc[2 * TR + lane];
                    mom[0] += a, mom[1] += b, mom[2] += c;
                    mom[3] = fmaf(a, a, mom[3]), mom[4] = fmaf(a, b, mom[4]), mom[5] = fmaf(a, c, mom[5]);
                    mom[6] = fmaf(b, b, mom[6]), mom[7] = fmaf(b, c, mom[7]), mom[8] = fmaf(c, c, mom[8]);
                }
            }
            if (!IN3) {
#pragma unroll
                for (int i = 0; i < 4; ++i) vout[i] = *reinterpret_cast<const float4 *>(Ts + (8 * i + (lane >> 3)) * 36 + (lane & 7) * 4);
            }
            }
            if (it == 1) SN_TL(2);
            if (more && SN_CBX_ABL != 4) {
                __bf16 *Zn = Lb + ((it + 1) & 1) * BUF;
                cbx_stage<CO, CI, TR, ZMODE, FULLR, NZ4, NP4, RZ1, PW>(g, nxt, ntic * TR, tid, Zn, Zn + 3 * ZPL, rz, rdy, rp, rag, rgs, k1, k2,
                                                                       k3, sc4, sh4, w3s);
                if (xthr) {
                    float *Xn = Xs + ((it + 1) & 1) * (3 * TR);
                    Xn[xslot[0]] = rx.x, Xn[xslot[1]] = rx.y, Xn[xslot[2]] = rx.z, Xn[xslot[3]] = rx.w;
                }
            }
            cloud = ncloud, tic = ntic;
            if (it == 1) SN_TL(3);
            __syncthreads();
            if (it == 1) SN_TL(4);
        }
        SN_TL(6);
        if (!IN3 && tile != tile0 && kh == 0) {
            const unsigned oso = (unsigned)(tile - tst) * (TR * GP * 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) buf_store4(vout[i], rs.dyprev, ovo, oso + i * (8 * GP * 4));
        }
        float *red = lds;  // [RB][NST][CI]   (every wave is past its last LDS read: barrier at the end of the loop)
        const float t0 = s0 + __shfl_xor(s0, 32), t1 = s1 + __shfl_xor(s1, 32);
        if (lane < 32 && kh == 0) {
            red[(rb * NST + 0) * CI + cb * 32 + lane] = t0;
            red[(rb * NST + 1) * CI + cb * 32 + lane] = t1;
        }
        if (IN3) {
            const float u0 = gx0 + __shfl_xor(gx0, 32), u1 = gx1 + __shfl_xor(gx1, 32), u2 = gx2 + __shfl_xor(gx2, 32);
            if (lane < 32) {
                red[(rb * NST + 2) * CI + cb * 32 + lane] = u0;
                red[(rb * NST + 3) * CI + cb * 32 + lane] = u1;
                red[(rb * NST + 4) * CI + cb * 32 + lane] = u2;
            }
            if (wave == 0) {
#pragma unroll
                for (int k = 0; k < 9; ++k) {
                    float m = mom[k];
#pragma unroll
                    for (int o = 32; o >= 1; o >>= 1) m += __shfl_xor(m, o);
                    if (lane == 0) red[RB * NST * CI + k] = m;
                }
            }
        }
    } else {
        // ---------------- weight-gradient waves ----------------------------------------------------------
        // (PW) this thread's part of the Zprev tile: 4 channels x NP4 rows, as the producer waves map theirs
        constexpr int PSTEPW = 256 / (CI / 4);
        const int tw = tid - 256;
        const int pc4w = (tw % (CI / 4)) * 4, prw = tw / (CI / 4);
        const unsigned pvow = (prw * GP + pc4w) * 4;
        const sn_rsrc rzp = make_rsrc(g.zprev, (unsigned)R * GP * 4);
        // two register sets: tile t + 2 is requested while tile t + 1 waits to be staged (the kernel moves ~48 KB per tile and CU;
        // with one tile in flight per CU the chip's HBM latency x bytes-in-flight product tops out near 3.6 TB/s)
        constexpr int PD = SN_CBX_PD;
        float4 rpw[PD][NP4];
        float4 scw = make_float4(0.f, 0.f, 0.f, 0.f), shw = scw;
        // RZ1: the rows' coordinates (12 bytes per row) in place of the Zprev tile, and the xyz layer's (w0, w1, w2, bias) of this
        // thread's four channels in registers (the weight-gradient waves of this layer hold one accumulator tile: room to spare)
        const sn_rsrc rsxw = make_rsrc(IN3 ? (const void *)g.xin : (const void *)g.dz.z, (unsigned)R * 12);
        const unsigned xvow = prw * 12;
        float4 w3w[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) w3w[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (PW && RZ1) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                w3w[j] = make_float4(g.w_in[(pc4w + j) * 3], g.w_in[(pc4w + j) * 3 + 1], g.w_in[(pc4w + j) * 3 + 2], g.b_in ? g.b_in[pc4w + j] : 0.f);
        }
        auto load_p = [&](int t, int s) __attribute__((always_inline)) {
#pragma unroll
            for (int q = 0; q < NP4; ++q)
                rpw[s][q] = RZ1 ? buf_load3(rsxw, xvow, (unsigned)t * (TR * 12) + q * (PSTEPW * 12))
                                : buf_load4(rzp, pvow, (unsigned)t * (TR * GP * 4) + q * (PSTEPW * GP * 4));
        };
        auto stage_p = [&](__bf16 *Pb, int s) __attribute__((always_inline)) {
#pragma unroll
            for (int q = 0; q < NP4; ++q) {
                float4 zp = rpw[s][q];
                if (RZ1) {  // the xyz kernel's own expression, bit for bit (cbx_stage's RZ1 branch)
#pragma clang fp contract(off)
                    const float x0 = rpw[s][q].x, x1 = rpw[s][q].y, x2 = rpw[s][q].z;
                    zp.x = fmaf(w3w[0].z, x2, fmaf(w3w[0].y, x1, w3w[0].x * x0)) + w3w[0].w;
                    zp.y = fmaf(w3w[1].z, x2, fmaf(w3w[1].y, x1, w3w[1].x * x0)) + w3w[1].w;
                    zp.z = fmaf(w3w[2].z, x2, fmaf(w3w[2].y, x1, w3w[2].x * x0)) + w3w[2].w;
                    zp.w = fmaf(w3w[3].z, x2, fmaf(w3w[3].y, x1, w3w[3].x * x0)) + w3w[3].w;
                }
                const float4 a = make_float4(relu_np(fmaf(zp.x, scw.x, shw.x)), relu_np(fmaf(zp.y, scw.y, shw.y)),
                                             relu_np(fmaf(zp.z, scw.z, shw.z)), relu_np(fmaf(zp.w, scw.w, shw.w)));
                stage_split_p<TR * LDP, LDP>(Pb, prw + q * PSTEPW, pc4w, a);
            }
        };
        if (PW) {
            load_p(tile0, 0);
            if (PD == 2) load_p(min(tile0 + tst, g.ntiles - 1), 1);
            scw = *reinterpret_cast<const float4 *>(g.scale_prev + pc4w);
            shw = *reinterpret_cast<const float4 *>(g.shift_prev + pc4w);
        }
        if (ZMODE == DZ_BN && g.acc_in != nullptr) {
            float *Ks = Tf;
            const int c = tid - 256;
            if (c < CO) {
                const BnBwd bb = g.bb_in;
                double su, sz;
                fx_get2<kFxShiftBwd>(g.acc_in, c, su, sz);
                const BnBwdOut o = bn_backward_coefs(bb.R, su, sz, bn_bwd_inputs(bb, CO, c));
                Ks[c] = o.k1, Ks[CO + c] = o.k2, Ks[2 * CO + c] = o.k3;
                if (blockIdx.x == 0) {
                    bb.dgamma[c] = o.dgamma, bb.dbeta[c] = o.dbeta;
                    if (bb.dbias) bb.dbias[c] = o.dbias;
                    if (bb.kcoef) bb.kcoef[c] = o.k1, bb.kcoef[CO + c] = o.k2, bb.kcoef[2 * CO + c] = o.k3;
                }
            }
            fx_clear_share(g.zero_ptr, g.zero_n, blockIdx.x, gridDim.x, tid - 256, 256);
            __syncthreads();
        }
        f32x16 accw[NWT];
#pragma unroll
        for (int n = 0; n < NWT; ++n)
#pragma unroll
            for (int e = 0; e < 16; ++e) accw[n][e] = 0.f;
        if (PW) stage_p(Lb + 3 * ZPL, 0);
        if (kSkew) load_p(tile0 + tst < tend ? tile0 + tst : tile0, 0);  // (slot A of the first tile stages it)
        __syncthreads();
        // transposing reads: this lane's row / channel offsets inside a [16 rows][32 channels] fragment block
        const int trr = 8 * h + ((lane & 15) >> 2), trc = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
        int tile = tile0;
        for (int it = 0; tile < tend; ++it, tile += tst) {
            const __bf16 *Zb = Lb + (it & 1) * BUF, *Pb = Zb + 3 * ZPL;
            const bool more = tile + tst < tend;
            if (kSkew) {
                // slot A: the next activation tile (requested one slot ago) into the other buffer, beside the data-gradient MFMAs
                if (more) stage_p(Lb + ((it + 1) & 1) * BUF + 3 * ZPL, 0);
                if (it == 1) SN_TL(2);
                __syncthreads();
                if (it == 1) SN_TL(3);
                load_p(tile + 2 * tst < tend ? tile + 2 * tst : tile, 0);
                __builtin_amdgcn_sched_barrier(0);
            } else if (PW) {
                // (PD == 2: set (it + 1) % 2 holds the next tile already; the set this tile's staging freed takes the one after)
                if (PD == 2) {
                    if (it & 1) load_p(min(tile + 2 * tst, g.ntiles - 1), 1); else load_p(min(tile + 2 * tst, g.ntiles - 1), 0);
                } else {
                    load_p(more ? tile + tst : tile, 0);
                }
                __builtin_amdgcn_sched_barrier(0);  // (the requests leave before the MFMAs, not next to the staging behind them)
            }
            // (SWZ: rows 16 kk + 8 h + 0..3 carry f = 2 h, the rows four below f = 2 h + 1)
            const __bf16 *ap = Zb + trr * LDZ + (S::SWZ ? (cob * 32 + trc) ^ ((2 * h) << 3) : cob * 32 + trc);
            const __bf16 *aph = Zb + (trr + 4) * LDZ + (S::SWZ ? (cob * 32 + trc) ^ ((2 * h + 1) << 3) : cob * 32 + trc);
            const __bf16 *bp = Pb + trr * LDP + trc;
            // fragments of k-step kk + 1 are requested BEFORE the MFMAs of k-step kk and the scheduler is held to that order: left
            // alone it interleaved a few transposing reads with a few MFMAs, each group behind its own wait (five to six exposed
            // LDS round trips per k-step; the weight-gradient waves took 2.7 us for 48 MFMAs that occupy the pipe for 0.64 us)
            bf16x8 fa[2][3], fb[2][3][NWT];
            auto frag_load = [&](int kk, int s) __attribute__((always_inline)) {
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    fa[s][p] = lds_tr8_2(ap + p * ZPL + kk * 16 * LDZ, aph + p * ZPL + kk * 16 * LDZ);
#pragma unroll
                    for (int n = 0; n < NWT; ++n) fb[s][p][n] = lds_tr8(bp + p * PPL + kk * 16 * LDP + ((q0 + n) % NCB) * 32, LDP);
                }
            };
            if (SN_CBX_ABL != 2) frag_load(0, 0);
#pragma unroll
            for (int kk = 0; kk < (SN_CBX_ABL == 2 ? 0 : KW); ++kk) {
                if (kk + 1 < KW) frag_load(kk + 1, (kk + 1) & 1);
                __builtin_amdgcn_sched_barrier(0);
                const bf16x8(&a)[3] = fa[kk & 1];
                const bf16x8(&b)[3][NWT] = fb[kk & 1];
#define SN_BX3_TERM(PA, PB) \
    _Pragma("unroll") for (int n = 0; n < NWT; ++n) accw[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA], b[PB][n], accw[n], 0, 0, 0)
                SN_BX3_TERM(0, 2);
                SN_BX3_TERM(2, 0);
                SN_BX3_TERM(1, 1);
                SN_BX3_TERM(0, 1);
                SN_BX3_TERM(1, 0);
                SN_BX3_TERM(0, 0);
#undef SN_BX3_TERM
                __builtin_amdgcn_sched_barrier(0);
            }
            if (it == 1) SN_TL(1);
            if (KS == 2) __syncthreads();  // (the dgrad waves' partial-tile hand-off)
            if (PW && !kSkew && more && SN_CBX_ABL != 4) {
                if (PD == 2 && !(it & 1)) stage_p(Lb + ((it + 1) & 1) * BUF + 3 * ZPL, 1);
                else stage_p(Lb + ((it + 1) & 1) * BUF + 3 * ZPL, 0);
            }
            __syncthreads();
            if (it == 1) SN_TL(4);
        }
        SN_TL(6);
        const int pld = g.part_ld > 0 ? g.part_ld : CI;
        float *P = g.part + (size_t)blockIdx.x * (g.part_wg_stride > 0 ? g.part_wg_stride : CO * CI);
        float *Tw = lds + RB * NST * CI + 16 + (wave - 4) * (32 * 36);  // behind the dgrad waves' statistics area
#pragma unroll
        for (int n = 0; n < NWT; ++n) {
            const int colb = ((q0 + n) % NCB) * 32;
#pragma unroll
            for (int e = 0; e < 16; ++e) Tw[frag_row(e, lane) * 36 + l31] = accw[n][e];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int rt = 8 * i + (lane >> 3);
                *reinterpret_cast<float4 *>(P + (size_t)(cob * 32 + rt) * pld + colb + (lane & 7) * 4) =
                    *reinterpret_cast<const float4 *>(Tw + rt * 36 + (lane & 7) * 4);
            }
        }
    }
    __syncthreads();
    if (tid < CI && DM != 1) {
        const float *red = lds;
        const int sld = g.stats_ld > 0 ? g.stats_ld : CI;
        float *st = g.stats + (size_t)blockIdx.x * (IN3 ? 6 : 2) * sld;
#pragma unroll
        for (int k = 0; k < NST; ++k) {
            float a = red[k * CI + tid];
            if (RB == 2) a += red[(NST + k) * CI + tid];
            if (!IN3 && g.acc_out)
                fx_add<kFxShiftBwd>(g.acc_out, blockIdx.x % kFxSlots, k, tid, a);
            else
                st[k * sld + tid] = a;
        }
        if (IN3) st[5 * CI + tid] = tid < 9 ? red[RB * NST * CI + tid] : 0.f;
    }
    SN_TL_DRAIN();
    SN_TL(7);
}

template <int ZMODE, int PMODE, bool VEC>
__device__ __forceinline__ void small_dgrad_body(const DgradArgs &g, int bx, float *lds)
{
    const int lane = threadIdx.x & 63, l31 = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int R = g.dz.rows, Co = g.w.co, Ci = g.w.ci;
    const int col = bx * 32 + l31;  // ci
    const bool colok = col < Ci;
    constexpr bool masked = PMODE == ACT_BN_RELU;
    // epilogue inputs first (previous layer's pre-BN activations at this lane's outputs, its BN coefficients)
    float zpv[16], sc = 0.f, sh = 0.f, pmean = 0.f, pinv = 0.f;
    if (masked) {
        const int cc = colok ? col : 0;
        sc = g.prev.scale[cc], sh = g.prev.shift[cc];
        if (g.bb.coef) pmean = g.bb.coef[2 * Ci + cc], pinv = g.bb.coef[3 * Ci + cc];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = frag_row(e, lane);
            zpv[e] = g.prev.z[(row < R && colok) ? (size_t)row * Ci + col : 0];
        }
    }
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    for (int k0 = wave * 2 * KP; k0 < Co; k0 += 4 * 2 * KP) {
        float a[KP], b[KP];
        const int kb = k0 + h * KP;
        if (VEC) {  // Co % 64 == 0
            const int rr = l31 < R ? l31 : 0;
            const float rmask = l31 < R ? 1.f : 0.f, cmask = colok ? 1.f : 0.f;
            const int cc = colok ? col : 0;
#pragma unroll
            for (int t = 0; t < KP; t += 4) {
                const float4 av = g.dz.template load_c4<true, ZMODE>(rr, kb + t);
                a[t] = av.x * rmask, a[t + 1] = av.y * rmask, a[t + 2] = av.z * rmask, a[t + 3] = av.w * rmask;
            }
#pragma unroll
            for (int t = 0; t < KP; ++t) b[t] = g.w.w[(size_t)(kb + t) * Ci + cc] * cmask;  // coalesced over lanes
        } else {
#pragma unroll
            for (int t = 0; t < KP; ++t) {
                const int k = kb + t;  // co
                a[t] = g.dz.template at<ZMODE>(l31, k);
                const bool ok = colok && k < Co;
                b[t] = g.w.w[ok ? (size_t)k * Ci + col : 0] * (ok ? 1.f : 0.f);
            }
        }
#ifdef SN_TIMELINE
        SN_TL_DRAIN();
        SN_TL(1);
#endif
#pragma unroll
        for (int t = 0; t < KP; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], b[t], acc, 0, 0, 0);
    }
    SN_TL(2);
    wave_sum_to_wave0(acc, lds);
    SN_TL(3);
    if (wave != 0) return;
    float s0 = 0.f, s1 = 0.f, s1c = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int row = frag_row(e, lane);
        if (row < R && colok) {
            float v = acc[e];
            if (masked) {
                const float zp = zpv[e];
                v = (fmaf(zp, sc, sh) > 0.f) ? v : 0.f;
                s0 += v;
                s1 += v * zp;
                s1c += v * (zp - pmean);  // centred: sum g (z - mean) without the cancellation of sum g z - mean sum g
            }
            g.dyprev[(size_t)row * Ci + col] = v;
        }
    }
    s0 += __shfl_xor(s0, 32);
    s1 += __shfl_xor(s1, 32);
    s1c += __shfl_xor(s1c, 32);
    if (masked && g.stats && lane < 32 && colok) g.stats[col] = s0, g.stats[Ci + col] = s1;
    if (masked && g.bb.coef && lane < 32 && colok) {  // bn_backward_channel on the prefetched mean / invstd
        const double scale = sc, mean = pmean, invstd = pinv, s = s0;
        const double dg = invstd * (double)s1c;
        g.bb.dgamma[col] = (float)dg;
        g.bb.dbeta[col] = (float)s;
        const double rinv = g.bb.R > 0 ? fast_rcp((double)g.bb.R) : 0.0;  // R <= 0: fixed statistics (see bn_backward_coefs)
        const float k1 = (float)scale, k2 = (float)(-scale * invstd * dg * rinv);
        const float k3 = (float)(scale * (invstd * mean * dg * rinv - s * rinv));
        g.bb.kcoef[col] = k1, g.bb.kcoef[Ci + col] = k2, g.bb.kcoef[2 * Ci + col] = k3;
        if (g.bb.dbias)
            g.bb.dbias[col] = (float)((double)k1 * s + (double)k2 * (double)g.bb.R * mean + (double)g.bb.R * (double)k3);
    }
}

template <int ZMODE, int PMODE, bool VEC>
__global__ void __launch_bounds__(256) small_dgrad_kernel(DgradArgs g)
{
    __shared__ float lds[3 * 16 * 64];
    small_dgrad_body<ZMODE, PMODE, VEC>(g, blockIdx.x, lds);
}

// dW[Co][Ci] (and db[Co] through the ones column) = dZ^T . act(prev), K = R <= 32: one wave per 32x32 output tile,
// no partials.
template <int ZMODE, int PMODE>
__device__ __forceinline__ void small_wgrad_body(const WgradArgs &g, float *__restrict__ dW, float *__restrict__ db,
                                                 int tiles_n, int ntiles, int bx)
{
    const int lane = threadIdx.x & 63, l31 = lane & 31, h = lane >> 5;
    const int tile = bx * 4 + (threadIdx.x >> 6);
    if (tile >= ntiles) return;
    const int m0 = (tile / tiles_n) * 32, n0 = (tile % tiles_n) * 32;
    const int Co = g.dz.ch, Ci = g.prev.ch, Ce = g.ncols, R = g.dz.rows;
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    float a[16], b[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const int r = h * 16 + t;
        a[t] = g.dz.template at<ZMODE>(r, m0 + l31);
        b[t] = g.prev.template at<PMODE>(r, n0 + l31);
    }
#pragma unroll
    for (int t = 0; t < 16; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], b[t], acc, 0, 0, 0);
    (void)R;
    if (Ce == Ci && (Ci & 31) == 0 && m0 + 32 <= Co) {
        // whole 32 x 32 tile of dW: transpose through LDS and store 16 bytes per lane (16 dword stores per wave cost ~58
        // issue cycles each -- the four waves of a workgroup were store-issue-bound)
        __shared__ float tw[4][32 * 36];
        float *T = tw[threadIdx.x >> 6];
#pragma unroll
        for (int e = 0; e < 16; ++e) T[frag_row(e, lane) * 36 + l31] = acc[e];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int rt = 8 * i + (lane >> 3);
            *reinterpret_cast<float4 *>(dW + (size_t)(m0 + rt) * Ci + n0 + (lane & 7) * 4) =
                *reinterpret_cast<const float4 *>(T + rt * 36 + (lane & 7) * 4);
        }
        return;
    }
    const int col = n0 + l31;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int row = m0 + frag_row(e, lane);
        if (row < Co && col < Ce) {
            if (col < Ci)
                dW[(size_t)row * Ci + col] = acc[e];
            else if (db)
                db[row] = acc[e];
        }
    }
}

template <int ZMODE, int PMODE>
__global__ void __launch_bounds__(256) small_wgrad_kernel(WgradArgs g, float *__restrict__ dW, float *__restrict__ db,
                                                          int tiles_n, int ntiles)
{
    small_wgrad_body<ZMODE, PMODE>(g, dW, db, tiles_n, ntiles, blockIdx.x);
}

// R <= 32 backward of one layer in ONE launch: workgroups [0, n_d) compute the data gradient (their epilogue also
// finishes the BatchNorm backward of the layer below), the rest the weight gradient.  Every launch on this chain costs
// a kernel boundary plus a cold first load (~4-5 us), whatever the amount of work.
template <int ZMODE, int PMODE, bool VEC>
__global__ void __launch_bounds__(256) small_bwd_kernel(DgradArgs d, WgradArgs w, float *__restrict__ dW,
                                                        float *__restrict__ db, int tiles_n, int ntiles, int n_d)
{
    __shared__ float lds[3 * 16 * 64];
    if ((int)blockIdx.x < n_d)
        small_dgrad_body<ZMODE, PMODE, VEC>(d, blockIdx.x, lds);
    else
        small_wgrad_body<ZMODE, PMODE>(w, dW, db, tiles_n, ntiles, blockIdx.x - n_d);
}

// ------------------------------------------------------------------------------------------------
// Layers on a few dozen to a few hundred rows (the FC head above 32 clouds: 33 .. ~500 rows, any count -- not only multiples of 64).
// The tile kernels treat these as small GEMMs of guarded 64 x 64 tiles walking K in dependent chunks plus a split-K weight gradient
// with its reduction launch (at 96 rows: 24 us per data gradient, 13 + 5 us per weight gradient -- the head's backward cost 190 of
// the step's 503 us; one row above the chain's 32 the step went from 0.19 to 0.32 ms).  Here the R <= 32 kernels run row block by
// row block in ONE launch per layer, every operand in flight at once:
//   data gradient    workgroup (column block, 64-row block) walks its two 32-row halves, mask / ReLU as small_dgrad_body; the
//                    BatchNorm-backward sums of the layer below leave as ONE partial per 64 rows ([ceil(R / 64)][2][Ci]: the
//                    layout bn_bwd_coef_kernel reduces)
//   weight gradient  one wave per 32 x 32 tile of dW walks ALL rows in 32-row steps (K = R): no partials, no reduction launch;
//                    the bias gradient is the ones column, as in small_wgrad_body
// ------------------------------------------------------------------------------------------------
template <int ZMODE, int PMODE, bool VEC>
__device__ __forceinline__ void rows_dgrad_body(const DgradArgs &g, int bx, int rb64, float *lds)
{
    // both 32-row halves of the 64-row block in ONE pass: the weight operand is fetched once and feeds two accumulators (walking
    // the halves one after the other doubled the launch's dependent latency: 14.3 us per 256 x 256 layer at 48 rows)
    const int lane = threadIdx.x & 63, l31 = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int R = g.dz.rows, Co = g.w.co, Ci = g.w.ci;
    const int col = bx * 32 + l31;  // ci
    const bool colok = col < Ci;
    const int cc = colok ? col : 0;
    constexpr bool masked = PMODE == ACT_BN_RELU;
    float sc = 0.f, sh = 0.f, pmean = 0.f, pinv = 0.f;
    if (masked) {
        sc = g.prev.scale[cc], sh = g.prev.shift[cc];
        // (R <= 64: the one 64-row block holds every row -- the host then asks for the BatchNorm-backward coefficients of the layer
        //  below right here, as small_dgrad_body delivers them, instead of a partial + a launch of its own)
        if (g.bb.coef) pmean = g.bb.coef[2 * Ci + cc], pinv = g.bb.coef[3 * Ci + cc];
    }
    const int rbase = rb64 * 64;
    const bool two = rbase + 32 < R;  // (uniform)
    float zpv[2][16];
    if (masked) {
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = rbase + 32 * s2 + frag_row(e, lane);
                zpv[s2][e] = g.prev.z[(row < R && colok) ? (size_t)row * Ci + col : 0];
            }
    }
    f32x16 acc[2];
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[0][e] = 0.f, acc[1][e] = 0.f;
    for (int k0 = wave * 2 * KP; k0 < Co; k0 += 4 * 2 * KP) {
        float a[2][KP], b[KP];
        const int kb = k0 + h * KP;
        if (VEC) {  // Co % 64 == 0
            const float cmask = colok ? 1.f : 0.f;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const int r = rbase + 32 * s2 + l31;
                const int rr = r < R ? r : 0;
                const float rmask = r < R ? 1.f : 0.f;
#pragma unroll
                for (int t = 0; t < KP; t += 4) {
                    const float4 av = g.dz.template load_c4<true, ZMODE>(rr, kb + t);
                    a[s2][t] = av.x * rmask, a[s2][t + 1] = av.y * rmask, a[s2][t + 2] = av.z * rmask, a[s2][t + 3] = av.w * rmask;
                }
            }
#pragma unroll
            for (int t = 0; t < KP; ++t) b[t] = g.w.w[(size_t)(kb + t) * Ci + cc] * cmask;
        } else {
#pragma unroll
            for (int t = 0; t < KP; ++t) {
                const int k = kb + t;
                a[0][t] = g.dz.template at<ZMODE>(rbase + l31, k);
                a[1][t] = g.dz.template at<ZMODE>(rbase + 32 + l31, k);
                const bool ok = colok && k < Co;
                b[t] = g.w.w[ok ? (size_t)k * Ci + col : 0] * (ok ? 1.f : 0.f);
            }
        }
#pragma unroll
        for (int t = 0; t < KP; ++t) acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0][t], b[t], acc[0], 0, 0, 0);
        if (two) {
#pragma unroll
            for (int t = 0; t < KP; ++t) acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1][t], b[t], acc[1], 0, 0, 0);
        }
    }
    wave_sum_to_wave0(acc[0], lds);
    if (two) wave_sum_to_wave0(acc[1], lds + 3 * 16 * 64);
    if (wave != 0) return;
    float s0 = 0.f, s1 = 0.f, s1c = 0.f;
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
        if (s2 == 1 && !two) break;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = rbase + 32 * s2 + frag_row(e, lane);
            if (row < R && colok) {
                float v = acc[s2][e];
                if (masked) {
                    const float zp = zpv[s2][e];
                    v = (fmaf(zp, sc, sh) > 0.f) ? v : 0.f;
                    s0 += v;
                    s1 += v * zp;
                    s1c += v * (zp - pmean);  // (centred, as small_dgrad_body)
                }
                g.dyprev[(size_t)row * Ci + col] = v;
            }
        }
    }
    if (!masked) return;
    s0 += __shfl_xor(s0, 32);
    s1 += __shfl_xor(s1, 32);
    s1c += __shfl_xor(s1c, 32);
    if (g.stats && lane < 32 && colok) {
        float *st = g.stats + (size_t)rb64 * 2 * Ci;
        st[col] = s0, st[Ci + col] = s1;
    }
    if (g.bb.coef && lane < 32 && colok) {  // bn_backward_channel on the prefetched mean / invstd (small_dgrad_body's epilogue)
        const double scale = sc, mean = pmean, invstd = pinv, s = s0;
        const double dg = invstd * (double)s1c;
        g.bb.dgamma[col] = (float)dg;
        g.bb.dbeta[col] = (float)s;
        const double rinv = g.bb.R > 0 ? fast_rcp((double)g.bb.R) : 0.0;
        const float k1 = (float)scale, k2 = (float)(-scale * invstd * dg * rinv);
        const float k3 = (float)(scale * (invstd * mean * dg * rinv - s * rinv));
        g.bb.kcoef[col] = k1, g.bb.kcoef[Ci + col] = k2, g.bb.kcoef[2 * Ci + col] = k3;
        if (g.bb.dbias)
            g.bb.dbias[col] = (float)((double)k1 * s + (double)k2 * (double)g.bb.R * mean + (double)g.bb.R * (double)k3);
    }
}

template <int ZMODE, int PMODE>
__device__ __forceinline__ void rows_wgrad_body(const WgradArgs &g, float *__restrict__ dW, float *__restrict__ db, int tiles_n,
                                                int ntiles, int bx)
{
    const int lane = threadIdx.x & 63, l31 = lane & 31, h = lane >> 5;
    const int tile = bx * 4 + (threadIdx.x >> 6);
    if (tile >= ntiles) return;
    const int m0 = (tile / tiles_n) * 32, n0 = (tile % tiles_n) * 32;
    const int Co = g.dz.ch, Ci = g.prev.ch, Ce = g.ncols, R = g.dz.rows;
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    for (int rb = 0; rb < R; rb += 32) {
        float a[16], b[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int r = rb + h * 16 + t;
            a[t] = g.dz.template at<ZMODE>(r, m0 + l31);
            b[t] = g.prev.template at<PMODE>(r, n0 + l31);
        }
#pragma unroll
        for (int t = 0; t < 16; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], b[t], acc, 0, 0, 0);
    }
    if (Ce == Ci && (Ci & 31) == 0 && m0 + 32 <= Co) {  // whole tile: 16-byte stores through LDS (small_wgrad_body)
        __shared__ float tw[4][32 * 36];
        float *T = tw[threadIdx.x >> 6];
#pragma unroll
        for (int e = 0; e < 16; ++e) T[frag_row(e, lane) * 36 + l31] = acc[e];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int rt = 8 * i + (lane >> 3);
            *reinterpret_cast<float4 *>(dW + (size_t)(m0 + rt) * Ci + n0 + (lane & 7) * 4) =
                *reinterpret_cast<const float4 *>(T + rt * 36 + (lane & 7) * 4);
        }
        return;
    }
    const int col = n0 + l31;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int row = m0 + frag_row(e, lane);
        if (row < Co && col < Ce) {
            if (col < Ci)
                dW[(size_t)row * Ci + col] = acc[e];
            else if (db)
                db[row] = acc[e];
        }
    }
}

template <int ZMODE, int PMODE, bool VEC>
__global__ void __launch_bounds__(256) rows_bwd_kernel(DgradArgs d, WgradArgs w, float *__restrict__ dW, float *__restrict__ db,
                                                       int tiles_n, int ntiles, int n_d, int ncb)
{
    __shared__ float lds[2 * 3 * 16 * 64];
    if ((int)blockIdx.x < n_d)
        rows_dgrad_body<ZMODE, PMODE, VEC>(d, blockIdx.x % ncb, blockIdx.x / ncb, lds);
    else
        rows_wgrad_body<ZMODE, PMODE>(w, dW, db, tiles_n, ntiles, blockIdx.x - n_d);
}

__global__ void __launch_bounds__(256) conv_in3_wgrad_kernel(int R, int Co, int rows_per_split, const float *__restrict__ x,
                                                             const float *__restrict__ dy, const float *__restrict__ z,
                                                             const float *__restrict__ kcoef, float *__restrict__ part)
{
    __shared__ float red[3][4][64];
    const int cl = threadIdx.x & 63, rq = threadIdx.x >> 6;
    const int co = blockIdx.y * 64 + cl;
    const bool ok = co < Co;
    const int cc = ok ? co : 0;
    const float k1 = kcoef[cc], k2 = kcoef[Co + cc], k3 = kcoef[2 * Co + cc];
    const int r0 = blockIdx.x * rows_per_split, r1 = min(R, r0 + rows_per_split);
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
#pragma unroll 4
    for (int r = r0 + rq; r < r1; r += 4) {
        const float d = fmaf(k1, dy[(size_t)r * Co + cc], fmaf(k2, z[(size_t)r * Co + cc], k3));
        a0 = fmaf(d, x[(size_t)r * 3 + 0], a0);
        a1 = fmaf(d, x[(size_t)r * 3 + 1], a1);
        a2 = fmaf(d, x[(size_t)r * 3 + 2], a2);
    }
    red[0][rq][cl] = a0, red[1][rq][cl] = a1, red[2][rq][cl] = a2;
    __syncthreads();
    if (rq == 0 && ok) {
        float *P = part + ((size_t)blockIdx.x * Co + co) * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) P[c] = (red[c][0][cl] + red[c][1][cl]) + (red[c][2][cl] + red[c][3][cl]);
    }
}

// partials [nsplit][Co][Ce] -> dW [Co][Ci], db [Co] (Ce = Ci + 1).  64 elements x 4 split-slices per workgroup;
// slices and the final 4-way sum run in a fixed order: deterministic.
__global__ void __launch_bounds__(1024) wgrad_reduce_kernel(int nsplit, int Co, int Ci, int Ce, const float *__restrict__ part,
                                                            float *__restrict__ dW, float *__restrict__ db)
{
    // 64 elements x 16 split-slices per workgroup, 8 independent loads in flight per thread (fixed-order sums)
    __shared__ float red[16][64];
    const int el = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + el;
    const size_t stride = (size_t)Co * Ce;
    float acc = 0.f;
    if (e < Co * Ce) {
        const float *p = part + e;
        int sp = sl;
        for (; sp + 7 * 16 < nsplit; sp += 8 * 16) {
            const float v0 = p[(size_t)sp * stride], v1 = p[(size_t)(sp + 16) * stride];
            const float v2 = p[(size_t)(sp + 32) * stride], v3 = p[(size_t)(sp + 48) * stride];
            const float v4 = p[(size_t)(sp + 64) * stride], v5 = p[(size_t)(sp + 80) * stride];
            const float v6 = p[(size_t)(sp + 96) * stride], v7 = p[(size_t)(sp + 112) * stride];
            acc += ((v0 + v1) + (v2 + v3)) + ((v4 + v5) + (v6 + v7));
        }
        for (; sp < nsplit; sp += 16) acc += p[(size_t)sp * stride];
    }
    red[sl][el] = acc;
    __syncthreads();
    if (sl == 0 && e < Co * Ce) {
        float tot = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) tot += red[q][el];
        const int o = e / Ce, i = e - o * Ce;
        if (i < Ci)
            dW[(size_t)o * Ci + i] = tot;
        else if (db)
            db[o] = tot;
    }
}

// backward: partial (sum dY, sum dY*Z) -> dgamma, dbeta and the per-channel dZ coefficients
//   dZ = scale * (dY - dbeta/R - zhat * dgamma/R),  zhat = (Z - mean) invstd
//      = k1 dY + k2 Z + k3
// dbias (gradient of the conv/linear bias in front of the BN) = sum_r dZ = k1 sum dY + k2 R mean + R k3 (== 0 up to rounding).
__global__ void __launch_bounds__(1024) bn_bwd_coef_kernel(int nblk, int C, const float *__restrict__ stats, BnBwd bb)
{
    double s, sz;
    const int c = blockIdx.x * kChan + (threadIdx.x & (kChan - 1));
    BnBwdIn in{};
    if (threadIdx.x < kChan && c < C) in = bn_bwd_inputs(bb, C, c);
    if (!partial_sums(nblk, C, stats, blockIdx.x, s, sz)) return;
    bn_backward_channel(bb, C, c, s, sz, in);
}

// dW[e] = sum over the nsplit partials part[sp][e], for the 256 elements of workgroup-block `blk`: 4 consecutive elements per
// thread (16-byte loads), 16 split-slices per workgroup, fixed-order sums.  nel % 4 == 0.
// Weight-gradient partials [nsplit][nel] -> their sum: a 1024-thread workgroup takes kRedElems consecutive elements, a thread
// sums partials sl, sl + NSL, ... of its 4 elements with kRedFlight 16-byte loads in flight, then the slices are added in index
// order (fixed summation order: run-to-run identical).  What matters is how many CONTIGUOUS bytes of one partial a workgroup
// touches -- measured on the conv stack's 32 MB (256 partials): 512 B 14.5 us, 1 KB 10.7, 2 KB 9.1, 4 KB 13.7 (too few
// workgroups), 8 KB 21.7; loads in flight (4 / 8 / 16) make no difference.
constexpr int kRedElems = 512, kRedFlight = 8;  // (nel % 4 == 0)
__device__ __forceinline__ void wgrad_reduce_block(int blk, int nel, int nsplit, const float *__restrict__ pp, float *__restrict__ out)
{
    constexpr int EL4 = kRedElems / 4, NSL = 1024 / EL4, NF = kRedFlight;
    __shared__ float4 red[NSL][EL4];
    const int el = threadIdx.x % EL4, sl = threadIdx.x / EL4;
    const int e = blk * kRedElems + el * 4;
    const size_t stride = (size_t)nel;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (e < nel) {
        const float *p = pp + e;
        for (int sp = sl; sp < nsplit; sp += NF * NSL) {
            float4 v[NF];
#pragma unroll
            for (int q = 0; q < NF; ++q) {  // (past the end: re-read the last partial, weighted 0 -- the loads stay unconditional)
                const int s2 = sp + NSL * q;
                v[q] = *reinterpret_cast<const float4 *>(p + (size_t)(s2 < nsplit ? s2 : nsplit - 1) * stride);
            }
#pragma unroll
            for (int q = 0; q < NF; ++q) {
                const float m = sp + NSL * q < nsplit ? 1.f : 0.f;
                acc.x = fmaf(v[q].x, m, acc.x), acc.y = fmaf(v[q].y, m, acc.y);
                acc.z = fmaf(v[q].z, m, acc.z), acc.w = fmaf(v[q].w, m, acc.w);
            }
        }
    }
    red[sl][el] = acc;
    __syncthreads();
    if (sl == 0 && e < nel) {
        float4 tot = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int q = 0; q < NSL; ++q) {
            const float4 v = red[q][el];
            tot.x += v.x, tot.y += v.y, tot.z += v.z, tot.w += v.w;
        }
        *reinterpret_cast<float4 *>(out + e) = tot;
    }
}

// wgrad_reduce of layer i and the BatchNorm backward coefficients of layer i-1 depend on the same launch (the combined
// backward kernel of layer i) and on nothing else: one launch for both.  Workgroups [0, nred) reduce, the rest do BN.
// db_dy / db_rows / db (optional, a plain-dZ layer with a bias -- the head's top layer): workgroups behind the BatchNorm ones sum
// the columns of dY (db_rows x Co) into db, 64 columns each, 16 row slices in fixed order.
__global__ void __launch_bounds__(1024) post_bwd_kernel(int nred, int nsplit, int Co, int Ci, const float *__restrict__ part,
                                                        float *__restrict__ dW, int nblk, int C,
                                                        const float *__restrict__ stats, BnBwd bb,
                                                        const float *__restrict__ db_dy = nullptr, int db_rows = 0,
                                                        float *__restrict__ db = nullptr)
{
    const int nbn = (C + kChan - 1) / kChan;
    if ((int)blockIdx.x >= nred + nbn) {
        __shared__ float dbred[16][64];
        const int cl = threadIdx.x & 63, sl = threadIdx.x >> 6;
        const int c = ((int)blockIdx.x - nred - nbn) * 64 + cl;
        float a = 0.f;
        if (c < Co)
            for (int r = sl; r < db_rows; r += 16) a += db_dy[(size_t)r * Co + c];
        dbred[sl][cl] = a;
        __syncthreads();
        if (sl == 0 && c < Co) {
            float t = 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) t += dbred[q][cl];
            db[c] = t;
        }
        return;
    }
    if ((int)blockIdx.x < nred) {
        if ((Co * Ci) % 4 == 0) {
            wgrad_reduce_block(blockIdx.x, Co * Ci, nsplit, part, dW);
            return;
        }
        // 64 elements x 16 split-slices per workgroup, 8 independent loads in flight per thread (fixed-order sums)
        __shared__ float red[16][64];
        const int el = threadIdx.x & 63, sl = threadIdx.x >> 6;
        const int e = blockIdx.x * 64 + el;
        const size_t stride = (size_t)Co * Ci;
        float acc = 0.f;
        if (e < Co * Ci) {
            const float *p = part + e;
            int sp = sl;
            for (; sp + 7 * 16 < nsplit; sp += 8 * 16) {
                const float v0 = p[(size_t)sp * stride], v1 = p[(size_t)(sp + 16) * stride];
                const float v2 = p[(size_t)(sp + 32) * stride], v3 = p[(size_t)(sp + 48) * stride];
                const float v4 = p[(size_t)(sp + 64) * stride], v5 = p[(size_t)(sp + 80) * stride];
                const float v6 = p[(size_t)(sp + 96) * stride], v7 = p[(size_t)(sp + 112) * stride];
                acc += ((v0 + v1) + (v2 + v3)) + ((v4 + v5) + (v6 + v7));
            }
            for (; sp < nsplit; sp += 16) acc += p[(size_t)sp * stride];
        }
        red[sl][el] = acc;
        __syncthreads();
        if (sl == 0 && e < Co * Ci) {
            float tot = 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) tot += red[q][el];
            dW[e] = tot;
        }
        return;
    }
    // BatchNorm backward coefficients, as bn_bwd_coef_kernel
    double s, sz;
    const int cblock = (int)blockIdx.x - nred;
    const int c = cblock * kChan + (threadIdx.x & (kChan - 1));
    BnBwdIn in{};
    if (threadIdx.x < kChan && c < C) in = bn_bwd_inputs(bb, C, c);
    if (!partial_sums(nblk, C, stats, cblock, s, sz)) return;
    bn_backward_channel(bb, C, c, s, sz, in);
}

// post_bwd_kernel behind the IN3 variant of conv_bwd_fused_kernel (statistics partials [nblk][6][C], see there): the BatchNorm
// workgroups also finish the weight gradient of the xyz input layer below, in closed form and in double:
//   dW_in[c][d] = k1 Gx[c][d] + k2 (sum_e W_in[c][e] Sxx[e][d] + b_in[c] Sx[d]) + k3 Sx[d]
// Several weight-gradient reductions in one launch (sn_conv_stack_backward: the partials of every conv layer are reduced
// at the end of the backward, not between its kernels).
struct MultiRed {
    int n;              // layers (<= 4); n == 0: single reduction described by the scalar arguments
    int first[5];       // first workgroup of layer i; first[n] = total
    const float *part[4];
    float *dW[4];
    int elems[4];       // Co * Ci
    long long *zero_ptr;
    int zero_n;
};

__global__ void __launch_bounds__(1024) post_bwd_in3_kernel(int nred, int nsplit, int Co, int Ci, const float *__restrict__ part,
                                                            float *__restrict__ dW, int nblk, int C,
                                                            const float *__restrict__ stats, BnBwd bb,
                                                            const float *__restrict__ W_in, const float *__restrict__ b_in,
                                                            float *__restrict__ dW_in, MultiRed mr, StepTail tail)
{
    kernarg_warm_for<0, int, int, int, int, const float *, float *, int, int, const float *, BnBwd, const float *, const float *, float *,
                     MultiRed, StepTail>();  // (-0.3 us: see sn_common.h; no gain in the GEMM kernels)
    // optional riders (engine path): the loss side's scalar tail in two extra workgroups at the end of the grid, and the
    // reset of its key table spread over the reduction workgroups
    if (tail.nparts > 0) {
        const int nbn = (C + kChan - 1) / kChan;
        if ((int)blockIdx.x == nred + nbn) {
            __shared__ float tred[4];
            sigma_grad_block(tail.nparts, tail.gsig, tail.temperature, tail.min_sigma, tail.grad_T, tail.grad_loss, tail.lmbda, tred);
            return;
        }
        if ((int)blockIdx.x == nred + nbn + 1) {
            if (threadIdx.x < 64) step_loss_keys_final(tail.kf, threadIdx.x);
            return;
        }
        if ((int)blockIdx.x < nred) {
            const long long per = (tail.kf.nkeys + nred - 1) / nred;
            const long long i0 = (long long)blockIdx.x * per, i1 = i0 + per < tail.kf.nkeys ? i0 + per : tail.kf.nkeys;
            for (long long i = i0 + threadIdx.x; i < i1; i += 1024) tail.kf.keys[i] = 0;
        }
    }
    if ((int)blockIdx.x < nred) {
        int blk = blockIdx.x, nel = Co * Ci;
        const float *pp = part;
        float *out = dW;
        if (mr.n > 0) {
            int li = 0;
#pragma unroll
            for (int q = 1; q < 4; ++q)
                if (q < mr.n && blk >= mr.first[q]) li = q;
            blk -= mr.first[li], nel = mr.elems[li], pp = mr.part[li], out = mr.dW[li];
            if (blockIdx.x == 0)
                for (int i = threadIdx.x; i < mr.zero_n; i += 1024) mr.zero_ptr[i] = 0;
        }
        wgrad_reduce_block(blk, nel, nsplit, pp, out);
        return;
    }
    // one memory round trip for everything this workgroup needs: thread (channel cl, slice sl) loads the five sums of its
    // channel from blocks sl, sl + 128, ...; threads 0 .. 575 also load the nine moments (9 x 64 slices)
    __shared__ double mred[9][64];
    __shared__ double mtot[9];
    __shared__ double red[5][kSlices][kChan];
    __shared__ double red2[5][16][kChan];
    const int cblock = (int)blockIdx.x - nred;
    const int cl = threadIdx.x & (kChan - 1), sl = threadIdx.x >> 3;
    const int c = cblock * kChan + cl;
    const size_t bs = (size_t)6 * C;
    BnBwdIn in{};
    float w0 = 0.f, w1 = 0.f, w2 = 0.f, bi = 0.f;
    if (threadIdx.x < kChan && c < C) {
        in = bn_bwd_inputs(bb, C, c);
        w0 = W_in[c * 3], w1 = W_in[c * 3 + 1], w2 = W_in[c * 3 + 2];
        if (b_in) bi = b_in[c];
    }
    double a[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
    if (c < C) {
        int b = sl;
        for (; b + kSlices < nblk; b += 2 * kSlices) {  // 10 independent loads in flight
            float u[5], v[5];
#pragma unroll
            for (int k = 0; k < 5; ++k) u[k] = stats[(size_t)b * bs + k * C + c], v[k] = stats[(size_t)(b + kSlices) * bs + k * C + c];
#pragma unroll
            for (int k = 0; k < 5; ++k) a[k] += (double)u[k] + (double)v[k];
        }
        for (; b < nblk; b += kSlices) {
#pragma unroll
            for (int k = 0; k < 5; ++k) a[k] += (double)stats[(size_t)b * bs + k * C + c];
        }
    }
    double ma = 0.0;
    if (threadIdx.x < 9 * 64) {
        const int m = threadIdx.x >> 6, ms = threadIdx.x & 63;
        for (int b = ms; b < nblk; b += 64) ma += (double)stats[(size_t)b * bs + 5 * C + m];
        mred[m][ms] = ma;
    }
#pragma unroll
    for (int k = 0; k < 5; ++k) red[k][sl][cl] = a[k];
    __syncthreads();
    if (sl < 16) {
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            double t = 0.0;
#pragma unroll
            for (int q = 0; q < 8; ++q) t += red[k][sl * 8 + q][cl];
            red2[k][sl][cl] = t;
        }
    } else if (threadIdx.x >= 512 && threadIdx.x < 512 + 9) {
        const int m = threadIdx.x - 512;
        double t = 0.0;
        for (int q = 0; q < 64; ++q) t += mred[m][q];
        mtot[m] = t;
    }
    __syncthreads();
    const bool own = sl == 0 && c < C;
    double s = 0.0, sz = 0.0, g0 = 0.0, g1 = 0.0, g2 = 0.0;
    if (own) {
#pragma unroll
        for (int q = 0; q < 16; ++q)
            s += red2[0][q][cl], sz += red2[1][q][cl], g0 += red2[2][q][cl], g1 += red2[3][q][cl], g2 += red2[4][q][cl];
    }
    if (!own) return;
    const float3 k = bn_backward_channel(bb, C, c, s, sz, in);
    const double Sx[3] = {mtot[0], mtot[1], mtot[2]};
    const double Sxx[3][3] = {{mtot[3], mtot[4], mtot[5]}, {mtot[4], mtot[6], mtot[7]}, {mtot[5], mtot[7], mtot[8]}};
    const double gx[3] = {g0, g1, g2};
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const double zx = (double)w0 * Sxx[0][d] + (double)w1 * Sxx[1][d] + (double)w2 * Sxx[2][d] + (double)bi * Sx[d];
        dW_in[c * 3 + d] = (float)((double)k.x * gx[d] + (double)k.y * zx + (double)k.z * Sx[d]);
    }
}

// backward of the pooling: gsel = g * [pooled > 0]; BN-backward partial sums of the last conv layer
// (one partial block: sum_b gsel, sum_b gsel * zsel), summed over b in ascending order.
__global__ void __launch_bounds__(1024) pool_bwd_kernel(int B, int C, const float *__restrict__ g,
                                                        const float *__restrict__ pooled, const float *__restrict__ zsel,
                                                        float *__restrict__ gsel, float *__restrict__ stats, BnBwd bb)
{
    // 64 channels x 16 batch slices per workgroup; slices and the final 16-way sum run in a fixed order
    __shared__ float red[2][16][64];
    const int cl = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    float s = 0.f, sz = 0.f;
    if (c < C) {
        int b = sl;
        for (; b + 3 * 16 < B; b += 4 * 16) {  // (four trips' loads in flight; the sums keep their ascending order)
            float pv[4], gv[4], zv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const size_t o = (size_t)(b + 16 * i) * C + c;
                pv[i] = pooled[o], gv[i] = g[o], zv[i] = zsel[o];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float v = pv[i] > 0.f ? gv[i] : 0.f;
                gsel[(size_t)(b + 16 * i) * C + c] = v;
                s += v;
                sz += v * zv[i];
            }
        }
        for (; b < B; b += 16) {
            const size_t o = (size_t)b * C + c;
            const float v = pooled[o] > 0.f ? g[o] : 0.f;
            gsel[o] = v;
            s += v;
            sz += v * zsel[o];
        }
    }
    red[0][sl][cl] = s, red[1][sl][cl] = sz;
    __syncthreads();
    if (sl == 0 && c < C) {
        float a0 = 0.f, a1 = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) a0 += red[0][q][cl], a1 += red[1][q][cl];
        if (stats) stats[c] = a0, stats[C + c] = a1;
        // the workgroup holds every cloud of its channels: the BatchNorm backward of the last conv layer completes here
        if (bb.coef) bn_backward_channel(bb, C, c, (double)a0, (double)a1);
    }
}

}  // namespace sn

static DzSrc make_dz(int mode, const float *dy, const float *z, const float *kcoef, const float *gsel, const int *argsel,
                     int rows, int ch, int npts)
{
    DzSrc d{};
    d.mode = mode, d.dy = dy, d.z = z, d.rows = rows, d.ch = ch, d.npts = npts > 0 ? npts : 1;
    d.k1 = kcoef, d.k2 = kcoef ? kcoef + ch : nullptr, d.k3 = kcoef ? kcoef + 2 * ch : nullptr;
    d.gsel = gsel, d.argsel = argsel;
    return d;
}

template <int ZMODE, int PMODE>
static void launch_dgrad(const DgradArgs &g, hipStream_t st)
{
    const int R = g.dz.rows, Ci = g.w.ci, Co = g.w.co;
    if (R <= 32) {
        if (Co % 64 == 0)
            hipLaunchKernelGGL((small_dgrad_kernel<ZMODE, PMODE, true>), dim3((Ci + 31) / 32), dim3(256), 0, st, g);
        else
            hipLaunchKernelGGL((small_dgrad_kernel<ZMODE, PMODE, false>), dim3((Ci + 31) / 32), dim3(256), 0, st, g);
    } else if (R > 64) {
        dim3 grid((R + TileBig::BM - 1) / TileBig::BM, (Ci + TileBig::BN - 1) / TileBig::BN);
        const bool full = R % TileBig::BM == 0 && Ci % TileBig::BN == 0 && Co % BK == 0;
        SN_LAUNCH_T(linear_dgrad_kernel, TileBig, full, grid, g, ZMODE, PMODE);
    } else {
        dim3 grid((R + TileSmall::BM - 1) / TileSmall::BM, (Ci + TileSmall::BN - 1) / TileSmall::BN);
        const bool full = R % TileSmall::BM == 0 && Ci % TileSmall::BN == 0 && Co % BK == 0;
        SN_LAUNCH_T(linear_dgrad_kernel, TileSmall, full, grid, g, ZMODE, PMODE);
    }
}

extern "C" int sn_linear_dgrad(int R, int Ci, int Co, int dz_mode, const float *dy, const float *z, const float *kcoef,
                               const float *gsel, const int *argsel, int npts, const float *W, const float *zprev,
                               const float *coef_prev, float *dyprev, float *stats, sn_stream_t stream)
{
    SN_REQUIRE(R >= 1 && Ci >= 1 && Co >= 1, "bad size");
    SN_REQUIRE(W && dyprev, "null pointer");
    SN_REQUIRE(dz_mode >= DZ_PLAIN && dz_mode <= DZ_POOL, "bad dz_mode");
    DgradArgs g{};
    g.dz = make_dz(dz_mode, dy, z, kcoef, gsel, argsel, R, Co, npts);
    g.w.w = W, g.w.co = Co, g.w.ci = Ci;
    g.prev = make_act(zprev, coef_prev, R, Ci);
    g.dyprev = dyprev, g.stats = stats;
    hipStream_t st = (hipStream_t)stream;
    const bool pm = coef_prev != nullptr;
    if (dz_mode == DZ_PLAIN) {
        if (pm) launch_dgrad<DZ_PLAIN, ACT_BN_RELU>(g, st); else launch_dgrad<DZ_PLAIN, ACT_NONE>(g, st);
    } else if (dz_mode == DZ_BN) {
        if (pm) launch_dgrad<DZ_BN, ACT_BN_RELU>(g, st); else launch_dgrad<DZ_BN, ACT_NONE>(g, st);
    } else {
        if (pm) launch_dgrad<DZ_POOL, ACT_BN_RELU>(g, st); else launch_dgrad<DZ_POOL, ACT_NONE>(g, st);
    }
    SN_LAUNCH_CHECK();
    return 0;
}

// ---- fused convolution backward (conv_bwd_fused_kernel): shapes, grid, launch -------------------------------------
static bool conv_bwd_fused_shape(int R, int Ci, int Co)
{
    if (R < 256) return false;
    if ((Ci == 64 && (Co == 64 || Co == 128)) || (Ci == 128 && Co == 128)) return true;
#if SN_BF16X3
    // 256 channels on one side (the reconstruction sampler's 128 -> 256 -> 128): two passes of the 128 x 128 kernel
    if ((Ci == 128 && Co == 256) || (Ci == 256 && Co == 128)) return true;
#endif
    return false;
}

// persistent workgroups: one per CU (each walks over ceil(tiles / groups) 64-row tiles)
static int conv_bwd_fused_groups(int R) { return std::min((R + 63) / 64, device_cus()); }

template <int CI, int CO, int ZMODE>
static void launch_conv_bwd_bx3_t(const ConvBwdArgs &a, int G, bool fullr, hipStream_t st)
{
    constexpr size_t lds = CbxShape<CI, CO>::LDS_BYTES;
    // (a refused request leaves the error text; the launch below then fails and the entry point's launch check reports it)
    static SnLdsAttr at, af;
    (void)sn_lds_attr(at, (const void *)conv_bwd_bx3_kernel<CI, CO, ZMODE, true>, lds, "conv_bwd_bx3_kernel");
    (void)sn_lds_attr(af, (const void *)conv_bwd_bx3_kernel<CI, CO, ZMODE, false>, lds, "conv_bwd_bx3_kernel");
    if (fullr)
        hipLaunchKernelGGL((conv_bwd_bx3_kernel<CI, CO, ZMODE, true>), dim3(G), dim3(512), lds, st, a);
    else
        hipLaunchKernelGGL((conv_bwd_bx3_kernel<CI, CO, ZMODE, false>), dim3(G), dim3(512), lds, st, a);
}

#if SN_BF16X3
// one pass of the 128 x 128 kernel over a half of a 256-channel side (see conv_bwd_bx3_kernel)
template <int ZMODE, int GZ, int GP, int GW, int DM>
static void launch_conv_bwd_bx3_half(const ConvBwdArgs &a, int G, bool fullr, hipStream_t st)
{
    constexpr size_t lds = CbxShape<128, 128>::LDS_BYTES;
    static SnLdsAttr at, af;
    (void)sn_lds_attr(at, (const void *)conv_bwd_bx3_kernel<128, 128, ZMODE, true, false, false, GZ, GP, GW, DM>, lds, "conv_bwd_bx3_kernel");
    (void)sn_lds_attr(af, (const void *)conv_bwd_bx3_kernel<128, 128, ZMODE, false, false, false, GZ, GP, GW, DM>, lds, "conv_bwd_bx3_kernel");
    if (fullr)
        hipLaunchKernelGGL((conv_bwd_bx3_kernel<128, 128, ZMODE, true, false, false, GZ, GP, GW, DM>), dim3(G), dim3(512), lds, st, a);
    else
        hipLaunchKernelGGL((conv_bwd_bx3_kernel<128, 128, ZMODE, false, false, false, GZ, GP, GW, DM>), dim3(G), dim3(512), lds, st, a);
}
#endif

template <int CI, int CO, int ZMODE>
static void launch_conv_bwd_fused_t(const ConvBwdArgs &a, int G, bool fullr, hipStream_t st)
{
#if SN_BF16X3
    launch_conv_bwd_bx3_t<CI, CO, ZMODE>(a, G, fullr, st);
    return;
#endif
    constexpr size_t lds = CbfShape<CI, CO>::LDS_BYTES;
    static SnLdsAttr at, af;  // more than 64 KB of dynamic LDS must be requested explicitly, per device
    (void)sn_lds_attr(at, (const void *)conv_bwd_fused_kernel<CI, CO, ZMODE, true>, lds, "conv_bwd_fused_kernel");
    (void)sn_lds_attr(af, (const void *)conv_bwd_fused_kernel<CI, CO, ZMODE, false>, lds, "conv_bwd_fused_kernel");
    if (fullr)
        hipLaunchKernelGGL((conv_bwd_fused_kernel<CI, CO, ZMODE, true>), dim3(G), dim3(512), lds, st, a);
    else
        hipLaunchKernelGGL((conv_bwd_fused_kernel<CI, CO, ZMODE, false>), dim3(G), dim3(512), lds, st, a);
}

// returns the number of workgroups (= partials in `stats` and `part`)
static int launch_conv_bwd_fused(int R, int Ci, int Co, int dz_mode, const float *dy, const float *z, const float *kcoef,
                                 const float *gsel, const int *argsel, int npts, const float *W, const float *zprev,
                                 const float *coef_prev, float *dyprev, float *stats, float *part, hipStream_t st,
                                 const ConvBwdArgs *fx = nullptr)
{
    ConvBwdArgs a{};
    if (fx) a.acc_in = fx->acc_in, a.bb_in = fx->bb_in, a.acc_out = fx->acc_out, a.zero_ptr = fx->zero_ptr, a.zero_n = fx->zero_n;
    a.dz.mode = dz_mode, a.dz.dy = dy, a.dz.z = z, a.dz.rows = R, a.dz.ch = Co, a.dz.npts = npts > 0 ? npts : 1;
    a.dz.k1 = kcoef, a.dz.k2 = kcoef ? kcoef + Co : nullptr, a.dz.k3 = kcoef ? kcoef + 2 * Co : nullptr;
    a.dz.gsel = gsel, a.dz.argsel = argsel;
    a.W = W, a.zprev = zprev, a.scale_prev = coef_prev, a.shift_prev = coef_prev + Ci;
    a.dyprev = dyprev, a.stats = stats, a.part = part;
#if SN_BF16X3
    const int TR = Co == 128 ? 32 : 64;  // CbxShape<Ci, Co>::TR
#else
    const int TR = (Ci == 128 && Co == 128) ? 32 : 64;  // CbfShape<Ci, Co>::TR
#endif
    a.ntiles = (R + TR - 1) / TR;
    const int G = conv_bwd_fused_groups(R);
    const bool fullr = R % TR == 0;
#if SN_BF16X3
    if (Co == 256 || Ci == 256) {
        a.ntiles = (R + 31) / 32;  // (CbxShape<128, 128>::TR)
        const bool f32 = R % 32 == 0;
        for (int hh = 0; hh < 2; ++hh) {
            ConvBwdArgs b = a;
            if (Co == 256) {  // halves of the output channels: dZ columns / W rows; the data gradient is the sum of the passes
                b.dz.z = z + 128 * hh, b.dz.dy = dy + 128 * hh, b.dz.ch = 128;
                b.dz.k1 = kcoef + 128 * hh, b.dz.k2 = kcoef + 256 + 128 * hh, b.dz.k3 = kcoef + 512 + 128 * hh;
                b.W = W + (size_t)128 * hh * 128;
                b.part = part + (size_t)128 * hh * 128, b.part_wg_stride = 256 * 128, b.part_ld = 128;
                b.dyacc = dyprev;  // (in place: a workgroup reads a tile's raw sums before it stores that tile's result)
                if (hh == 0) launch_conv_bwd_bx3_half<DZ_BN, 256, 128, 128, 1>(b, G, f32, st);
                else launch_conv_bwd_bx3_half<DZ_BN, 256, 128, 128, 2>(b, G, f32, st);
            } else {  // halves of the input channels: W / Zprev / dYprev columns, independent
                b.W = W + 128 * hh, b.zprev = zprev + 128 * hh, b.dyprev = dyprev + 128 * hh;
                b.scale_prev = coef_prev + 128 * hh, b.shift_prev = coef_prev + 256 + 128 * hh;
                b.stats = stats + 128 * hh, b.stats_ld = 256;
                b.part = part + 128 * hh, b.part_wg_stride = 128 * 256, b.part_ld = 256;
                if (dz_mode == DZ_BN) launch_conv_bwd_bx3_half<DZ_BN, 128, 256, 256, 0>(b, G, f32, st);
                else launch_conv_bwd_bx3_half<DZ_POOL, 128, 256, 256, 0>(b, G, f32, st);
            }
        }
        return G;
    }
#endif
#define SN_CBF(CI_, CO_)                                                                       \
    do {                                                                                       \
        if (dz_mode == DZ_BN) launch_conv_bwd_fused_t<CI_, CO_, DZ_BN>(a, G, fullr, st);        \
        else launch_conv_bwd_fused_t<CI_, CO_, DZ_POOL>(a, G, fullr, st);                       \
    } while (0)
    if (Ci == 64 && Co == 64) SN_CBF(64, 64);
    else if (Ci == 64 && Co == 128) SN_CBF(64, 128);
    else SN_CBF(128, 128);
#undef SN_CBF
    return G;
}

static bool conv_bwd_fused_ok(int R, int Ci, int Co, int dz_mode, int npts, const float *coef_prev, const float *kcoef,
                              const float *db)
{
    return !db && coef_prev && kcoef && conv_bwd_fused_shape(R, Ci, Co) &&
           (dz_mode == DZ_BN || (dz_mode == DZ_POOL && npts > 0 && npts % 64 == 0 && Co != 256));  // (256 outputs: DZ_BN passes only)
}

extern "C" int sn_linear_wgrad_splits(int R, int Ci, int Co, int with_bias)
{
    if (R <= 32) return 1;  // small path writes dW directly (scratch unused)
    if (!with_bias && conv_bwd_fused_shape(R, Ci, Co)) return conv_bwd_fused_groups(R);  // one partial per workgroup
    const int ncols = Ci + (with_bias ? 1 : 0);
    const int tiles = ((Co + TileW::BM - 1) / TileW::BM) * ((ncols + TileW::BN - 1) / TileW::BN);
    const int want = std::max(1, 512 / tiles);                       // aim at ~2 workgroups per CU
    const int maxsplit = std::max(1, (R + 2 * BK - 1) / (2 * BK));   // at least 128 rows per split
    return std::max(1, std::min(want, maxsplit));
}

template <int ZMODE, int PMODE>
static void launch_wgrad(WgradArgs &g, int R, int Ci, int Co, int with_bias, float *dW, float *db, hipStream_t st)
{
    if (R <= 32) {  // K = R fits one MFMA K range: one wave per 32x32 output tile, written directly
        const int tm = (Co + 31) / 32, tn = (g.ncols + 31) / 32;
        hipLaunchKernelGGL((small_wgrad_kernel<ZMODE, PMODE>), dim3((tm * tn + 3) / 4), dim3(256), 0, st, g, dW, db, tn,
                           tm * tn);
        return;
    }
    const int nsplit = sn_linear_wgrad_splits(R, Ci, Co, with_bias);
    int rps = (R + nsplit - 1) / nsplit;
    rps = ((rps + BK - 1) / BK) * BK;
    g.rows_per_split = rps;
    if (ZMODE == DZ_BN && PMODE == ACT_NONE && Ci == 3 && !with_bias) {  // xyz input layer
        hipLaunchKernelGGL(conv_in3_wgrad_kernel, dim3(nsplit, (Co + 63) / 64), dim3(256), 0, st, R, Co, rps, g.prev.z, g.dz.dy,
                           g.dz.z, g.dz.k1, g.part);
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((Co * 3 + 63) / 64), dim3(1024), 0, st, nsplit, Co, 3, 3, g.part, dW, db);
        return;
    }
    dim3 grid((Co + TileW::BM - 1) / TileW::BM, (g.ncols + TileW::BN - 1) / TileW::BN, nsplit);
    // fast path: every split covers whole K chunks of in-range rows and whole output tiles
    const bool full = !with_bias && Co % TileW::BM == 0 && Ci % TileW::BN == 0 && R % rps == 0;
    SN_LAUNCH_T(linear_wgrad_kernel, TileW, full, grid, g, ZMODE, PMODE);
    const int tot = Co * g.ncols;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((tot + 63) / 64), dim3(1024), 0, st, nsplit, Co, Ci, g.ncols, g.part, dW, db);
}

// part: scratch of sn_linear_wgrad_splits(...) * Co * (Ci + with_bias) floats.  db may be NULL (no bias column).
extern "C" int sn_linear_wgrad(int R, int Ci, int Co, int dz_mode, const float *dy, const float *z, const float *kcoef,
                               const float *gsel, const int *argsel, int npts, const float *aprev,
                               const float *coef_prev, float *part, float *dW, float *db, sn_stream_t stream)
{
    SN_REQUIRE(R >= 1 && Ci >= 1 && Co >= 1, "bad size");
    SN_REQUIRE(aprev && part && dW, "null pointer");
    SN_REQUIRE(dz_mode >= DZ_PLAIN && dz_mode <= DZ_POOL, "bad dz_mode");
    const int with_bias = db != nullptr;
    WgradArgs g{};
    g.dz = make_dz(dz_mode, dy, z, kcoef, gsel, argsel, R, Co, npts);
    g.prev = make_act(aprev, coef_prev, R, Ci, with_bias ? Ci : -1);
    g.ncols = Ci + with_bias;
    g.part = part;
    hipStream_t st = (hipStream_t)stream;
    const bool pm = coef_prev != nullptr;
    if (dz_mode == DZ_PLAIN) {
        if (pm) launch_wgrad<DZ_PLAIN, ACT_BN_RELU>(g, R, Ci, Co, with_bias, dW, db, st);
        else launch_wgrad<DZ_PLAIN, ACT_NONE>(g, R, Ci, Co, with_bias, dW, db, st);
    } else if (dz_mode == DZ_BN) {
        if (pm) launch_wgrad<DZ_BN, ACT_BN_RELU>(g, R, Ci, Co, with_bias, dW, db, st);
        else launch_wgrad<DZ_BN, ACT_NONE>(g, R, Ci, Co, with_bias, dW, db, st);
    } else {
        if (pm) launch_wgrad<DZ_POOL, ACT_BN_RELU>(g, R, Ci, Co, with_bias, dW, db, st);
        else launch_wgrad<DZ_POOL, ACT_NONE>(g, R, Ci, Co, with_bias, dW, db, st);
    }
    SN_LAUNCH_CHECK();
    return 0;
}

// The fused convolution backward on its own (what sn_linear_backward / sn_layer_backward launch first for 64 / 128-channel
// layers): dYprev, BatchNorm-backward partial sums [G][2][Ci] and dW partials [G][Co][Ci] with
// G = sn_linear_wgrad_splits(R, Ci, Co, 0); the caller reduces the partials (sn_linear_backward does).  Returns
// SN_ERR_UNSUPPORTED for shapes the fused kernel does not serve.
extern "C" int sn_conv_backward_partials(int R, int Ci, int Co, int dz_mode, const float *dy, const float *z, const float *kcoef,
                                         const float *gsel, const int *argsel, int npts, const float *W, const float *zprev,
                                         const float *coef_prev, float *dyprev, float *stats, float *part, sn_stream_t stream)
{
    SN_REQUIRE(R >= 1 && Ci >= 1 && Co >= 1, "bad size");
    SN_REQUIRE(W && zprev && dyprev && part && stats && z && kcoef && coef_prev, "null pointer");
    SN_REQUIRE((dz_mode == DZ_BN && dy) || (dz_mode == DZ_POOL && gsel && argsel), "bad dz_mode / missing gradient source");
    if (!conv_bwd_fused_ok(R, Ci, Co, dz_mode, npts, coef_prev, kcoef, nullptr))
        return sn_set_error(SN_ERR_UNSUPPORTED, "sn_conv_backward_partials: shape not served by the fused kernel");
    launch_conv_bwd_fused(R, Ci, Co, dz_mode, dy, z, kcoef, gsel, argsel, npts, W, zprev, coef_prev, dyprev, stats, part,
                          (hipStream_t)stream);
    SN_LAUNCH_CHECK();
    return 0;
}

// dgrad + wgrad of one layer.  Arguments as sn_linear_dgrad / sn_linear_wgrad (aprev == zprev: the previous layer's
// pre-BN activations, or the raw input when coef_prev == NULL).  One launch on the fast path, else the two kernels.
// ---- rows_bwd_kernel: one launch per layer on 33 .. 512 rows (see the kernel) ----------------------------------------------
static bool rows_bwd_shape(int R, int dz_mode)
{
    // same-box sweep of the whole step (ms per step, tile kernels -> this launch): 48 rows 0.346 -> 0.284, 96 0.509 -> 0.405,
    // 128 0.470 -> 0.450, 192 0.666 -> 0.594, 50 x 2048 points of configs[3] 0.842 -> 0.774; equal at 64 and 320, slower at 256 /
    // 384 / 512 (0.710 -> 0.737 at 256: a weight-gradient wave walks all R rows -- the split-K tiles win from there).  So: every row count the tile kernels' fast path
    // (whole 64-row blocks) does not serve, up to 512, and the multiples of 64 up to 192 (at 64 rows and below the launch also
    // delivers the BatchNorm-backward coefficients of the layer below: one launch per layer)
    if (R <= 32 || dz_mode == DZ_POOL) return false;
    return R % 64 != 0 ? R <= 512 : R <= 192;
}

// -> the number of BatchNorm-backward partial blocks written to stats ([ceil(R / 64)][2][Ci], when coef_prev and stats)
static int launch_rows_bwd(int R, int Ci, int Co, int dz_mode, const float *dy, const float *z, const float *kcoef, const float *W,
                           const float *zprev, const float *coef_prev, float *dyprev, float *stats, float *dW, float *db, hipStream_t st,
                           const BnBwd *finalize = nullptr)
{
    DgradArgs g{};
    if (finalize) g.bb = *finalize;  // (R <= 64 only: one row block holds every row)
    g.dz = make_dz(dz_mode, dy, z, kcoef, nullptr, nullptr, R, Co, 1);
    g.w.w = W, g.w.co = Co, g.w.ci = Ci;
    g.prev = make_act(zprev, coef_prev, R, Ci);
    g.dyprev = dyprev, g.stats = stats;
    WgradArgs wg{};
    wg.dz = g.dz;
    wg.prev = make_act(zprev, coef_prev, R, Ci, db ? Ci : -1);
    wg.ncols = Ci + (db ? 1 : 0);
    const int tm = (Co + 31) / 32, tn = (wg.ncols + 31) / 32, ntiles = tm * tn;
    const int ncb = (Ci + 31) / 32, nrb = (R + 63) / 64, n_d = ncb * nrb, n_w = (ntiles + 3) / 4;
    const dim3 grid(n_d + n_w), block(256);
    const bool pm = coef_prev != nullptr, vec = Co % 64 == 0;
#define SN_RB(ZM, PM)                                                                                                        \
    do {                                                                                                                     \
        if (vec)                                                                                                             \
            hipLaunchKernelGGL((rows_bwd_kernel<ZM, PM, true>), grid, block, 0, st, g, wg, dW, db, tn, ntiles, n_d, ncb);     \
        else                                                                                                                 \
            hipLaunchKernelGGL((rows_bwd_kernel<ZM, PM, false>), grid, block, 0, st, g, wg, dW, db, tn, ntiles, n_d, ncb);    \
    } while (0)
    if (dz_mode == DZ_PLAIN) {
        if (pm) SN_RB(DZ_PLAIN, ACT_BN_RELU); else SN_RB(DZ_PLAIN, ACT_NONE);
    } else {
        if (pm) SN_RB(DZ_BN, ACT_BN_RELU); else SN_RB(DZ_BN, ACT_NONE);
    }
#undef SN_RB
    return nrb;
}

extern "C" int sn_linear_backward(int R, int Ci, int Co, int dz_mode, const float *dy, const float *z, const float *kcoef,
                                  const float *gsel, const int *argsel, int npts, const float *W, const float *zprev,
                                  const float *coef_prev, float *dyprev, float *stats, float *part, float *dW,
                                  sn_stream_t stream)
{
    SN_REQUIRE(R >= 1 && Ci >= 1 && Co >= 1, "bad size");
    SN_REQUIRE(W && zprev && dyprev && part && dW, "null pointer");
    SN_REQUIRE(dz_mode >= DZ_PLAIN && dz_mode <= DZ_POOL, "bad dz_mode");
    if (conv_bwd_fused_ok(R, Ci, Co, dz_mode, npts, coef_prev, kcoef, nullptr)) {
        SN_REQUIRE(stats && z && (dz_mode != DZ_BN || dy) && (dz_mode != DZ_POOL || (gsel && argsel)), "null pointer");
        const int G = launch_conv_bwd_fused(R, Ci, Co, dz_mode, dy, z, kcoef, gsel, argsel, npts, W, zprev, coef_prev, dyprev,
                                            stats, part, (hipStream_t)stream);
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((Co * Ci + 63) / 64), dim3(1024), 0, (hipStream_t)stream, G, Co, Ci, Ci,
                           part, dW, nullptr);
        SN_LAUNCH_CHECK();
        return 0;
    }
    if (rows_bwd_shape(R, dz_mode)) {
        SN_REQUIRE((dz_mode == DZ_PLAIN || z) && dy, "null pointer");
        launch_rows_bwd(R, Ci, Co, dz_mode, dy, z, kcoef, W, zprev, coef_prev, dyprev, stats, dW, nullptr, (hipStream_t)stream);
        SN_LAUNCH_CHECK();
        return 0;
    }
    const int nsplit = sn_linear_wgrad_splits(R, Ci, Co, 0);
    int rps = (R + nsplit - 1) / nsplit;
    rps = ((rps + BK - 1) / BK) * BK;
    const bool fast = R > 64 && coef_prev && dz_mode != DZ_PLAIN && R % TileBig::BM == 0 && Ci % TileBig::BN == 0 &&
                      Co % BK == 0 && Co % TileW::BM == 0 && Ci % TileW::BN == 0 && R % rps == 0;
    if (!fast) {
        int rc = sn_linear_wgrad(R, Ci, Co, dz_mode, dy, z, kcoef, gsel, argsel, npts, zprev, coef_prev, part, dW, nullptr, stream);
        if (rc) return rc;
        return sn_linear_dgrad(R, Ci, Co, dz_mode, dy, z, kcoef, gsel, argsel, npts, W, zprev, coef_prev, dyprev, stats, stream);
    }
    hipStream_t st = (hipStream_t)stream;
    DgradArgs d{};
    d.dz = make_dz(dz_mode, dy, z, kcoef, gsel, argsel, R, Co, npts);
    d.w.w = W, d.w.co = Co, d.w.ci = Ci;
    d.prev = make_act(zprev, coef_prev, R, Ci);
    d.dyprev = dyprev, d.stats = stats;
    WgradArgs w{};
    w.dz = d.dz;
    w.prev = make_act(zprev, coef_prev, R, Ci, -1);
    w.ncols = Ci, w.part = part, w.rows_per_split = rps;
    const int wgx = Co / TileW::BM, wgy = Ci / TileW::BN, n_w = wgx * wgy * nsplit;
    const int dgx = R / TileBig::BM, n_d = dgx * (Ci / TileBig::BN);
    const dim3 grid(n_w + n_d);
    const size_t lds = shaped_lds(std::max(lds_bytes<TileBig>(), lds_bytes<TileW>()), grid);
    static_assert(TileBig::THREADS == TileW::THREADS, "combined backward kernel needs one workgroup size");
    if (dz_mode == DZ_BN)
        hipLaunchKernelGGL((linear_bwd_kernel<TileBig, DZ_BN, ACT_BN_RELU>), grid, dim3(TileBig::THREADS), lds, st, d, w, n_w, wgx, wgy, dgx);
    else
        hipLaunchKernelGGL((linear_bwd_kernel<TileBig, DZ_POOL, ACT_BN_RELU>), grid, dim3(TileBig::THREADS), lds, st, d, w, n_w, wgx, wgy, dgx);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((Co * Ci + 63) / 64), dim3(1024), 0, st, nsplit, Co, Ci, Ci, part, dW, nullptr);
    SN_LAUNCH_CHECK();
    return 0;
}

// Backward of one layer INCLUDING the BatchNorm backward coefficients of the layer below it:
//   dW (and db when db != NULL: bias column, plain dz only), dYprev, and -- when the layer below has a BatchNorm
//   (coef_prev != NULL) -- its dgamma / dbeta / dbias / kcoef[3][Ci].
// R <= 32: two launches (register-resident wgrad; dgrad whose epilogue finishes the BatchNorm backward);
// large R fast path: the combined dgrad+wgrad launch + one launch for (wgrad reduce | BatchNorm coefficients).
extern "C" int sn_layer_backward(int R, int Ci, int Co, int dz_mode, const float *dy, const float *z, const float *kcoef,
                                 const float *gsel, const int *argsel, int npts, const float *W, const float *zprev,
                                 const float *coef_prev, float *dyprev, float *stats, float *part, float *dW, float *db,
                                 float *prev_dgamma, float *prev_dbeta, float *prev_dbias, float *prev_kcoef,
                                 long long prev_bn_rows, sn_stream_t stream)
{
    SN_REQUIRE(R >= 1 && Ci >= 1 && Co >= 1, "bad size");
    SN_REQUIRE(W && zprev && dyprev && dW, "null pointer");
    SN_REQUIRE(dz_mode >= DZ_PLAIN && dz_mode <= DZ_POOL, "bad dz_mode");
    SN_REQUIRE(!coef_prev || ((stats || R <= 32) && prev_dgamma && prev_dbeta && prev_kcoef), "previous-layer BatchNorm outputs missing");
    SN_REQUIRE(prev_bn_rows <= 0 || R <= 32, "prev_bn_rows > 0 applies to the register-resident (R <= 32) path only");
    hipStream_t st = (hipStream_t)stream;
    // prev_bn_rows: rows the BatchNorm of the layer below averaged over when they are not this layer's R -- the FC head's
    // first layer sits on the max-pool of the last conv layer: zprev = the pooled pre-BN values (B rows), its BatchNorm saw
    // B * N rows; the ReLU mask / sums of the dgrad epilogue are then exactly the pooling backward
    // prev_bn_rows < 0: that BatchNorm ran on fixed (running) statistics -- eval-mode backward, dZ = scale * dY
    const BnBwd bb{coef_prev, prev_dgamma, prev_dbeta, prev_dbias, prev_kcoef,
                   prev_bn_rows > 0 ? prev_bn_rows : (prev_bn_rows < 0 ? -1ll : (long long)R)};
    if (R <= 32) {
        DgradArgs g{};
        g.dz = make_dz(dz_mode, dy, z, kcoef, gsel, argsel, R, Co, npts);
        g.w.w = W, g.w.co = Co, g.w.ci = Ci;
        g.prev = make_act(zprev, coef_prev, R, Ci);
        g.dyprev = dyprev, g.stats = nullptr;
        if (coef_prev) g.bb = bb;
        WgradArgs wg{};
        wg.dz = g.dz;
        wg.prev = make_act(zprev, coef_prev, R, Ci, db ? Ci : -1);
        wg.ncols = Ci + (db ? 1 : 0);
        const int tm = (Co + 31) / 32, tn = (wg.ncols + 31) / 32, ntiles = tm * tn;
        const int n_d = (Ci + 31) / 32, n_w = (ntiles + 3) / 4;
        const dim3 grid(n_d + n_w), block(256);
        const bool pm = coef_prev != nullptr, vec = Co % 64 == 0;
#define SN_SB(ZM, PM)                                                                                                   \
    do {                                                                                                                \
        if (vec)                                                                                                        \
            hipLaunchKernelGGL((small_bwd_kernel<ZM, PM, true>), grid, block, 0, st, g, wg, dW, db, tn, ntiles, n_d);    \
        else                                                                                                            \
            hipLaunchKernelGGL((small_bwd_kernel<ZM, PM, false>), grid, block, 0, st, g, wg, dW, db, tn, ntiles, n_d);   \
    } while (0)
        if (dz_mode == DZ_PLAIN) {
            if (pm) SN_SB(DZ_PLAIN, ACT_BN_RELU); else SN_SB(DZ_PLAIN, ACT_NONE);
        } else if (dz_mode == DZ_BN) {
            if (pm) SN_SB(DZ_BN, ACT_BN_RELU); else SN_SB(DZ_BN, ACT_NONE);
        } else {
            if (pm) SN_SB(DZ_POOL, ACT_BN_RELU); else SN_SB(DZ_POOL, ACT_NONE);
        }
#undef SN_SB
        SN_LAUNCH_CHECK();
        return 0;
    }
    SN_REQUIRE(part, "scratch missing");
    if (conv_bwd_fused_ok(R, Ci, Co, dz_mode, npts, coef_prev, kcoef, db)) {
        SN_REQUIRE(z && (dz_mode != DZ_BN || dy) && (dz_mode != DZ_POOL || (gsel && argsel)), "null pointer");
        const int G = launch_conv_bwd_fused(R, Ci, Co, dz_mode, dy, z, kcoef, gsel, argsel, npts, W, zprev, coef_prev, dyprev,
                                            stats, part, st);
        const int nred = (Co * Ci) % 4 == 0 ? (Co * Ci + kRedElems - 1) / kRedElems : (Co * Ci + 63) / 64;
        hipLaunchKernelGGL(post_bwd_kernel, dim3(nred + (Ci + kChan - 1) / kChan), dim3(1024), 0, st, nred, G, Co, Ci, part, dW,
                           G, Ci, stats, bb);
        SN_LAUNCH_CHECK();
        return 0;
    }
    if (rows_bwd_shape(R, dz_mode)) {
        SN_REQUIRE((dz_mode == DZ_PLAIN || z) && dy, "null pointer");
        const bool fin = coef_prev && R <= 64;  // the BatchNorm-backward coefficients of the layer below inside the launch
        const int nb = launch_rows_bwd(R, Ci, Co, dz_mode, dy, z, kcoef, W, zprev, coef_prev, dyprev, fin ? nullptr : stats, dW, db, st,
                                       fin ? &bb : nullptr);
        if (coef_prev && !fin)
            hipLaunchKernelGGL(bn_bwd_coef_kernel, dim3((Ci + kChan - 1) / kChan), dim3(1024), 0, st, nb, Ci, stats, bb);
        SN_LAUNCH_CHECK();
        return 0;
    }
    const int nsplit = sn_linear_wgrad_splits(R, Ci, Co, db ? 1 : 0);
    int rps = (R + nsplit - 1) / nsplit;
    rps = ((rps + BK - 1) / BK) * BK;
    // the head's top layer above 32 rows (dZ = dY as given, bias gradient wanted): the combined launch as well -- the bias gradient
    // is the column sums of dY, taken by extra workgroups of the closing launch instead of a ones column in the weight-gradient
    // tiles (four launches before: wgrad with the bias column, its reduction, dgrad, BatchNorm coefficients; 30 -> 17 us at 512 rows)
    const bool plain_top = dz_mode == DZ_PLAIN && dy != nullptr;
    const bool fast = (plain_top || (!db && dz_mode != DZ_PLAIN)) && coef_prev && R % TileBig::BM == 0 && Ci % TileBig::BN == 0 &&
                      Co % BK == 0 && Co % TileW::BM == 0 && Ci % TileW::BN == 0 && R % rps == 0;
    const int nblk = sn_linear_stats_blocks(R);
    if (!fast) {
        int rc = sn_linear_wgrad(R, Ci, Co, dz_mode, dy, z, kcoef, gsel, argsel, npts, zprev, coef_prev, part, dW, db, stream);
        if (rc) return rc;
        rc = sn_linear_dgrad(R, Ci, Co, dz_mode, dy, z, kcoef, gsel, argsel, npts, W, zprev, coef_prev, dyprev, stats, stream);
        if (rc) return rc;
        if (coef_prev)
            hipLaunchKernelGGL(bn_bwd_coef_kernel, dim3((Ci + kChan - 1) / kChan), dim3(1024), 0, st, nblk, Ci, stats, bb);
        SN_LAUNCH_CHECK();
        return 0;
    }
    DgradArgs d{};
    d.dz = make_dz(dz_mode, dy, z, kcoef, gsel, argsel, R, Co, npts);
    d.w.w = W, d.w.co = Co, d.w.ci = Ci;
    d.prev = make_act(zprev, coef_prev, R, Ci);
    d.dyprev = dyprev, d.stats = stats;
    WgradArgs w{};
    w.dz = d.dz;
    w.prev = make_act(zprev, coef_prev, R, Ci, -1);
    w.ncols = Ci, w.part = part, w.rows_per_split = rps;
    const int wgx = Co / TileW::BM, wgy = Ci / TileW::BN, n_w = wgx * wgy * nsplit;
    const int dgx = R / TileBig::BM, n_d = dgx * (Ci / TileBig::BN);
    const dim3 grid(n_w + n_d);
    const size_t lds = shaped_lds(std::max(lds_bytes<TileBig>(), lds_bytes<TileW>()), grid);
    if (dz_mode == DZ_BN)
        hipLaunchKernelGGL((linear_bwd_kernel<TileBig, DZ_BN, ACT_BN_RELU>), grid, dim3(TileBig::THREADS), lds, st, d, w, n_w, wgx, wgy, dgx);
    else if (dz_mode == DZ_POOL)
        hipLaunchKernelGGL((linear_bwd_kernel<TileBig, DZ_POOL, ACT_BN_RELU>), grid, dim3(TileBig::THREADS), lds, st, d, w, n_w, wgx, wgy, dgx);
    else
        hipLaunchKernelGGL((linear_bwd_kernel<TileBig, DZ_PLAIN, ACT_BN_RELU>), grid, dim3(TileBig::THREADS), lds, st, d, w, n_w, wgx, wgy, dgx);
    const int nred = (Co * Ci) % 4 == 0 ? (Co * Ci + kRedElems - 1) / kRedElems : (Co * Ci + 63) / 64;
    const int ndb = db ? (Co + 63) / 64 : 0;
    // (the statistics partials THIS route fills: one per TileBig row block -- at R == 64 sn_linear_stats_blocks counts TileSmall
    //  blocks, and summing that many read an unwritten block: wrong dgamma / dbeta / dbias of the layer below at exactly 64 rows)
    hipLaunchKernelGGL(post_bwd_kernel, dim3(nred + (Ci + kChan - 1) / kChan + ndb), dim3(1024), 0, st, nred, nsplit, Co, Ci, part, dW, dgx,
                       Ci, stats, bb, dy, R, db);
    SN_LAUNCH_CHECK();
    return 0;
}

// sn_layer_backward for the layer that sits on the xyz input layer (Ci -> Co on top of 3 -> Ci): the fused backward
// also accumulates what the input layer's weight gradient needs (conv_bwd_fused_kernel IN3, post_bwd_in3_kernel), so that
// gradient costs no pass of its own over dYprev -- which is then not even written (8 MB less traffic).  stats: sn_layer_backward_in3_stats_floats(R, Ci, Co) floats (0 = shape
// not supported: use sn_layer_backward + sn_linear_wgrad).
extern "C" long long sn_layer_backward_in3_stats_floats(int R, int Ci, int Co)
{
    if (R < 1 || !(Ci == 64 && Co == 64) || !conv_bwd_fused_shape(R, Ci, Co)) return 0;
    return (long long)conv_bwd_fused_groups(R) * 6 * Ci;
}

static void launch_conv_bwd_in3(int R, const float *dy, const float *z, const float *kcoef, const float *W, const float *zprev,
                                const float *coef_prev, float *stats, float *part, const float *x_in, hipStream_t st,
                                const ConvBwdArgs *fx = nullptr, const float *w_in = nullptr, const float *b_in = nullptr)
{
    constexpr int Ci = 64, Co = 64;
    ConvBwdArgs a{};
    a.w_in = w_in, a.b_in = b_in;  // (zprev == NULL: the xyz layer's parameters, Zprev is rebuilt from x_in)
    if (fx) a.acc_in = fx->acc_in, a.bb_in = fx->bb_in, a.acc_out = nullptr, a.zero_ptr = fx->zero_ptr, a.zero_n = fx->zero_n;
    a.dz.mode = DZ_BN, a.dz.dy = dy, a.dz.z = z, a.dz.rows = R, a.dz.ch = Co, a.dz.npts = 1;
    a.dz.k1 = kcoef, a.dz.k2 = kcoef ? kcoef + Co : nullptr, a.dz.k3 = kcoef ? kcoef + 2 * Co : nullptr;
    a.W = W, a.zprev = zprev, a.scale_prev = coef_prev, a.shift_prev = coef_prev + Ci;
    a.dyprev = nullptr, a.stats = stats, a.part = part, a.xin = x_in;  // dYprev is not materialised: nothing reads it
    constexpr int TR = CbfShape<64, 64>::TR;
    static_assert(TR == CbxShape<64, 64>::TR, "same tiling in both kernels");
    a.ntiles = (R + TR - 1) / TR;
    const int G = conv_bwd_fused_groups(R);
#if SN_BF16X3
#define SN_CBF_IN3 conv_bwd_bx3_kernel
    constexpr size_t lds = CbxShape<64, 64>::LDS_BYTES_IN3;
    if (!zprev) {  // Zprev rebuilt from the cloud (the forward did not materialise it)
        static SnLdsAttr rt, rf;
        (void)sn_lds_attr(rt, (const void *)conv_bwd_bx3_kernel<64, 64, DZ_BN, true, true, true>, lds, "conv_bwd_bx3_kernel");
        (void)sn_lds_attr(rf, (const void *)conv_bwd_bx3_kernel<64, 64, DZ_BN, false, true, true>, lds, "conv_bwd_bx3_kernel");
        if (R % TR == 0)
            hipLaunchKernelGGL((conv_bwd_bx3_kernel<64, 64, DZ_BN, true, true, true>), dim3(G), dim3(512), lds, st, a);
        else
            hipLaunchKernelGGL((conv_bwd_bx3_kernel<64, 64, DZ_BN, false, true, true>), dim3(G), dim3(512), lds, st, a);
        return;
    }
#else
#define SN_CBF_IN3 conv_bwd_fused_kernel
    constexpr size_t lds = CbfShape<64, 64>::LDS_BYTES_IN3;
#endif
    static SnLdsAttr at, af;
    (void)sn_lds_attr(at, (const void *)SN_CBF_IN3<64, 64, DZ_BN, true, true>, lds, "conv_bwd (xyz layer below)");
    (void)sn_lds_attr(af, (const void *)SN_CBF_IN3<64, 64, DZ_BN, false, true>, lds, "conv_bwd (xyz layer below)");
    if (R % TR == 0)
        hipLaunchKernelGGL((SN_CBF_IN3<64, 64, DZ_BN, true, true>), dim3(G), dim3(512), lds, st, a);
    else
        hipLaunchKernelGGL((SN_CBF_IN3<64, 64, DZ_BN, false, true>), dim3(G), dim3(512), lds, st, a);
#undef SN_CBF_IN3
}

extern "C" int sn_layer_backward_in3(int R, int Ci, int Co, const float *dy, const float *z, const float *kcoef, const float *W,
                                     const float *zprev, const float *coef_prev, float *stats, float *part,
                                     float *dW, float *prev_dgamma, float *prev_dbeta, float *prev_dbias, float *prev_kcoef,
                                     const float *x_in, const float *W_in, const float *b_in, float *dW_in, sn_stream_t stream)
{
    SN_REQUIRE(sn_layer_backward_in3_stats_floats(R, Ci, Co) > 0, "shape not supported by the input-layer variant");
    SN_REQUIRE(dy && z && kcoef && W && zprev && coef_prev && stats && part && dW, "null pointer");
    SN_REQUIRE(prev_dgamma && prev_dbeta && prev_kcoef && x_in && W_in && dW_in, "null pointer");
    hipStream_t st = (hipStream_t)stream;
    launch_conv_bwd_in3(R, dy, z, kcoef, W, zprev, coef_prev, stats, part, x_in, st);
    const int G = conv_bwd_fused_groups(R);
    const BnBwd bb{coef_prev, prev_dgamma, prev_dbeta, prev_dbias, prev_kcoef, (long long)R};
    const int nred = (Co * Ci + kRedElems - 1) / kRedElems;
    hipLaunchKernelGGL(post_bwd_in3_kernel, dim3(nred + (Ci + kChan - 1) / kChan), dim3(1024), 0, st, nred, G, Co, Ci, part, dW, G,
                       Ci, stats, bb, W_in, b_in, dW_in, MultiRed{}, StepTail{});
    SN_LAUNCH_CHECK();
    return 0;
}

// Backward of the whole conv stack (the mirror of sn_conv_stack_forward_bn) in nlayers launches: one fused dgrad + wgrad
// kernel per GEMM layer, top first, and ONE closing kernel that reduces every layer's weight-gradient partials and
// finishes the xyz layer (BatchNorm backward + closed-form weight gradient).  Between the kernels the BatchNorm-backward
// sums travel as fixed-point atomics (acc): each kernel derives its own layer's dZ coefficients in its prologue.
// Inputs: x (B*N,3); per layer W, z (pre-BN outputs), coef (4,C); gsel / argsel (B,Cn) + kcoef_top (3,Cn): the pooled
// gradient at the selected points and the top BatchNorm's dZ coefficients (from the FC side: sn_layer_backward with
// prev_bn_rows, or sn_pool_backward_bn).  Outputs: dW per layer; dgamma / dbeta / dbias for layers 0 .. nlayers-2.
// acc: sn_conv_stack_acc_elems(nlayers) long long, zero before the first call (left zero); scratch: see _scratch_floats.
// step_tail (optional): blob of sn_step_tail_bytes() bytes filled by sn_sampler_step_loss_keys(..., deferred_tail): the loss
// side's sigma gradient / loss value / key-table reset ride in the closing kernel instead of a launch of their own.
static bool conv_stack_backward_ok(int B, int N, int nlayers, const int *ch)
{
    if (!sn_conv_stack_forward_supported(B, N, nlayers, ch) || nlayers < 3 || nlayers > 5) return false;
    const int R = B * N;
    if (ch[1] != 64 || ch[2] != 64 || R < 256) return false;
    for (int l = 1; l < nlayers; ++l)  // (above 128 channels: the forward's two-block accumulator layout, two-pass backward kernels)
        if (ch[l] > 128 || ch[l + 1] > 128 || !conv_bwd_fused_shape(R, ch[l], ch[l + 1])) return false;
    return true;
}

extern "C" long long sn_conv_stack_backward_scratch_floats(int B, int N, int nlayers, const int *channels)
{
    if (!conv_stack_backward_ok(B, N, nlayers, channels)) return 0;
    const long long R = (long long)B * N, G = conv_bwd_fused_groups((int)R);
    long long n = 0;
    for (int l = 1; l < nlayers; ++l) n += G * channels[l] * channels[l + 1];  // weight-gradient partials
    for (int l = 2; l < nlayers; ++l) n += R * channels[l];                     // dY of layers 1 .. nlayers-2 ... (dY_{l-1})
    n += G * 6 * 64 + 3 * 64;                                                   // xyz-layer statistics partials, its kcoef
    return n;
}

extern "C" int sn_conv_stack_backward(int B, int N, int nlayers, const int *channels, const float *x, const float *const *W,
                                      const float *bias0, const float *const *z, const float *const *coef, const float *gsel,
                                      const int *argsel, const float *kcoef_top, long long *acc, float *scratch,
                                      float *const *dW, float *const *dgamma, float *const *dbeta, float *const *dbias,
                                      const void *step_tail, sn_stream_t stream)
{
    if (!conv_stack_backward_ok(B, N, nlayers, channels))
        return sn_set_error(SN_ERR_UNSUPPORTED, "sn_conv_stack_backward: shape not supported (use the per-layer entries)");
    SN_REQUIRE(x && W && z && coef && gsel && argsel && kcoef_top && acc && scratch && dW && dgamma && dbeta && dbias, "null pointer");
    for (int l = 0; l < nlayers; ++l) SN_REQUIRE(W[l] && (z[l] || l == 0) && coef[l] && dW[l], "null pointer");
    for (int l = 0; l + 1 < nlayers; ++l) SN_REQUIRE(dgamma[l] && dbeta[l] && dbias[l], "null pointer");
#if !SN_BF16X3
    SN_REQUIRE(z[0], "z[0] == NULL needs the split-bf16 build");
#endif
    hipStream_t st = (hipStream_t)stream;
    const int R = B * N, G = conv_bwd_fused_groups(R);
    const int *ch = channels;
    float *part[5] = {}, *dy[5] = {};
    float *p = scratch;
    for (int l = 1; l < nlayers; ++l) part[l] = p, p += (size_t)G * ch[l] * ch[l + 1];
    for (int l = 2; l < nlayers; ++l) dy[l - 1] = p, p += (size_t)R * ch[l];  // dy[l-1]: gradient at layer l-1's activations
    float *stats0 = p;
    p += (size_t)G * 6 * 64;
    float *kcoef0 = p;
    auto accb = [&](int l) { return acc + (size_t)l * kFxLayer; };
    for (int L = nlayers - 1; L >= 1; --L) {
        const int Ci = ch[L], Co = ch[L + 1];
        ConvBwdArgs fx{};
        const bool top = L == nlayers - 1;
        if (!top) {
            fx.acc_in = accb(L);
            fx.bb_in = BnBwd{coef[L], dgamma[L], dbeta[L], dbias[L], nullptr, (long long)R};
            if (L + 1 <= nlayers - 2) fx.zero_ptr = accb(L + 1), fx.zero_n = kFxLayer;
        }
        if (L >= 2) {
            fx.acc_out = accb(L - 1);
            launch_conv_bwd_fused(R, Ci, Co, top ? DZ_POOL : DZ_BN, top ? nullptr : dy[L], z[L], top ? kcoef_top : nullptr,
                                  top ? gsel : nullptr, top ? argsel : nullptr, N, W[L], z[L - 1], coef[L - 1], dy[L - 1], nullptr,
                                  part[L], st, &fx);
        } else {
            launch_conv_bwd_in3(R, dy[1], z[1], nullptr, W[1], z[0], coef[0], stats0, part[1], x, st, &fx, W[0], bias0);
        }
    }
    MultiRed mr{};
    mr.n = nlayers - 1;
    int nb = 0;
    for (int l = 1; l < nlayers; ++l) {
        mr.first[l - 1] = nb, mr.part[l - 1] = part[l], mr.dW[l - 1] = dW[l], mr.elems[l - 1] = ch[l] * ch[l + 1];
        nb += (ch[l] * ch[l + 1] + kRedElems - 1) / kRedElems;
    }
    mr.first[nlayers - 1] = nb;
    mr.zero_ptr = accb(1), mr.zero_n = kFxLayer;
    const BnBwd bb0{coef[0], dgamma[0], dbeta[0], dbias[0], kcoef0, (long long)R};
    StepTail tail{};
    if (step_tail) memcpy(&tail, step_tail, sizeof(tail));  // blob filled by sn_sampler_step_loss_keys (sn_step_tail_bytes())
    hipLaunchKernelGGL(post_bwd_in3_kernel, dim3(nb + (64 + kChan - 1) / kChan + (tail.nparts > 0 ? 2 : 0)), dim3(1024), 0, st, nb, G,
                       64, 64, part[1], dW[1], G, 64, stats0, bb0, W[0], bias0, dW[0], mr, tail);
    SN_LAUNCH_CHECK();
    return 0;
}

extern "C" int sn_bn_backward_coef(int nblk, int C, long long R, const float *stats, const float *coef, float *dgamma,
                                   float *dbeta, float *dbias, float *kcoef, sn_stream_t stream)
{
    SN_REQUIRE(nblk >= 1 && C >= 1 && R >= 1 && stats && coef && dgamma && dbeta && kcoef, "bad argument");
    const BnBwd bb{coef, dgamma, dbeta, dbias, kcoef, R};
    hipLaunchKernelGGL(bn_bwd_coef_kernel, dim3((C + kChan - 1) / kChan), dim3(1024), 0, (hipStream_t)stream, nblk, C, stats, bb);
    SN_LAUNCH_CHECK();
    return 0;
}

extern "C" int sn_pool_backward(int B, int C, const float *g, const float *pooled, const float *zsel, float *gsel,
                                float *stats, sn_stream_t stream)
{
    SN_REQUIRE(B >= 1 && C >= 1 && g && pooled && zsel && gsel && stats, "bad argument");
    hipLaunchKernelGGL(pool_bwd_kernel, dim3((C + 63) / 64), dim3(1024), 0, (hipStream_t)stream, B, C, g, pooled, zsel,
                       gsel, stats, BnBwd{});
    SN_LAUNCH_CHECK();
    return 0;
}

// sn_pool_backward + sn_bn_backward_coef of the last conv layer in one launch (R = rows the BatchNorm saw = B * N)
extern "C" int sn_pool_backward_bn(int B, int C, long long R, const float *g, const float *pooled, const float *zsel,
                                   float *gsel, const float *coef, float *dgamma, float *dbeta, float *dbias, float *kcoef,
                                   sn_stream_t stream)
{
    SN_REQUIRE(B >= 1 && C >= 1 && R != 0 && g && pooled && zsel && gsel && coef && dgamma && dbeta && kcoef, "bad argument");
    const BnBwd bb{coef, dgamma, dbeta, dbias, kcoef, R};  // R < 0: fixed statistics (eval-mode backward)
    hipLaunchKernelGGL(pool_bwd_kernel, dim3((C + 63) / 64), dim3(1024), 0, (hipStream_t)stream, B, C, g, pooled, zsel,
                       gsel, (float *)nullptr, bb);
    SN_LAUNCH_CHECK();
    return 0;
}
