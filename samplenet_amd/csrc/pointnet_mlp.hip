// pointnet_mlp.hip -- the PointNet feature extractor + FC head of SampleNet on gfx950 matrix cores.
//
// Reference: registration/src/samplenet.py:40-59 (parameters) and :90-104 (forward):
//   5 x [Conv1d(k=1) -> BatchNorm1d -> ReLU]  3->64->64->64->128->bottleneck over B*N points,
//   max over N, 3 x [Linear -> BatchNorm1d -> ReLU] 256, Linear -> 3*M.
// A 1x1 convolution over points is a GEMM with R = B*N rows; a Linear layer is the same GEMM
// with R = B rows.  Every layer (forward, data-gradient, weight-gradient) runs on ONE tiled
// GEMM core built on v_mfma_f32_32x32x2_f32 (exact fp32: each product rounded once, fp32
// accumulate -- bitwise a k-ordered fmaf chain), with the surrounding elementwise work fused in:
//
//   forward  Z = relu(bn_prev(Zprev)) . W^T + b     BN-apply + ReLU of the PREVIOUS layer fused into
//                                                   the A-operand load; bias + per-channel
//                                                   sum / sum-of-squares (BatchNorm batch statistics)
//                                                   fused into the epilogue (deterministic two-stage
//                                                   reduction, no atomics)
//   dgrad    dY_prev = mask(relu) . (dZ . W)        BN-backward of THIS layer fused into the A load
//                                                   (dZ = k1*dY + k2*Z + k3 per channel), ReLU mask
//                                                   + BN-backward statistics of the previous layer
//                                                   fused into the epilogue
//   wgrad    dW|db = dZ^T . [relu(bn(Zprev)) | 1]   both operands rebuilt on the fly from the stored
//                                                   pre-BN activations; bias gradient = extra column
//                                                   of ones; split over R, partials reduced in order
// Only the pre-BN activations Z_i are ever written to HBM (once) and read back (forward: once,
// backward: by dgrad and wgrad).
//
// GEMM core: block tile BM x BN, K chunks of 32 staged through LDS k-major ([k][m], so an MFMA
// fragment is one conflict-free ds_read_b32: lanes 0-31 read 32 consecutive floats of row k,
// lanes 32-63 of row k+1), next chunk prefetched into registers while the current one is in
// the matrix pipe.
//
// This unit: the FORWARD side -- layer GEMM tiles, the persistent weight-stationary forward, the R <= 32 layer kernels, the xyz
// input layer, BatchNorm finalisation and max-pool kernels, the conv-stack forward entry points.  The backward side lives in
// pointnet_mlp_backward.hip, the FC head chains in fc_chain.hip, the task network (PCRNet) kernels in task_network.hip; shared
// code: mlp_device.h (device), mlp_host.h (launch helpers).
#include "mlp_host.h"

namespace sn {

// KT > 0 (statistics-chain path): the input width, known at compile time -- both operands are fetched whole, up front
// PLANES: FwdArgs::wplanes holds the weights pre-split (statistics-chain path)
// (two 512-thread workgroups per CU need <= 128 registers: T::THREADS / 128 waves per SIMD at least)
template <class T, bool FULL, int AMODE, int KT = 0, bool PLANES = false, bool IN3A = false>
__global__ void __launch_bounds__(T::THREADS) __attribute__((amdgpu_waves_per_eu(T::THREADS / 128, 8))) linear_fwd_kernel(FwdArgs g)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    SN_TL(0);
    SN_TL_ID(0);
    const int row0 = blockIdx.x * T::BM, col0 = blockIdx.y * T::BN;
    const int R = g.a.rows, Co = g.w.co, Ci = g.w.ci;
    f32x16 acc[T::TM][T::TN];
#pragma unroll
    for (int i = 0; i < T::TM; ++i)
#pragma unroll
        for (int j = 0; j < T::TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    const ActSrc a = g.a;
    const WSrc w = g.w;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wave / T::WC, wc = wave % T::WC;
    // epilogue inputs are fetched BEFORE the GEMM: a load issued in the epilogue would expose a full memory latency
    float biasv[T::TN];
#pragma unroll
    for (int j = 0; j < T::TN; ++j) {
        const int col = col0 + (wc * T::TN + j) * 32 + (lane & 31);
        biasv[j] = g.bias ? g.bias[(FULL || col < Co) ? col : 0] : 0.f;
    }
    if (AMODE == ACT_BN_RELU_FX) {
        // finalise the input's BatchNorm from its fixed-point sums (same arithmetic as bn_finalize_channel).  Order of the
        // memory operations: (1) the accumulator words of this thread's channel, (2) the first chunk of both GEMM operands,
        // RAW -- then the coefficients are computed while (2) is still in flight (loads return in order: waiting for (1)
        // does not wait for (2)).  Chaining the two round trips cost ~2 us per layer.
        static_assert(FULL || AMODE != ACT_BN_RELU_FX, "fixed-point statistics chain: full tiles only");
        float *cf = lds + T::LDS_FLOATS;  // [2][Ci] scale | shift
        const BnFwd bp = g.bn_prev;
        const bool first = blockIdx.x == 0 && blockIdx.y == 0;
        const int c = threadIdx.x;  // THREADS >= Ci (<= 256)
        long long lo0[kFxSlots], lo1[kFxSlots], ha = 0, hb = 0, poison = 0;
        float bg = 0.f, bb = 0.f, brm = 0.f, brv = 0.f;
        if (c < Ci) {
            // (KT > 128: the input layer's sums fill two accumulator blocks back to back, channel c in block c >> 7)
            const long long *ai = KT > kFxRow ? g.acc_in + (c / kFxRow) * kFxLayer : g.acc_in;
            const int cl = KT > kFxRow ? c % kFxRow : c;
            poison = ai[kFxPoison];
#pragma unroll
            for (int q = 0; q < kFxSlots; ++q) lo0[q] = ai[(q * 2 + 0) * kFxRow + cl], lo1[q] = ai[(q * 2 + 1) * kFxRow + cl];
            ha = ai[kFxHi + cl], hb = ai[kFxHi + kFxRow + cl];
            bg = bp.gamma[c], bb = bp.beta[c];
            if (first && bp.running_mean) brm = bp.running_mean[c], brv = bp.running_var[c];
        }
        static_assert(!IN3A || (PLANES && KT > 0 && SN_BF16X3), "IN3A: statistics-chain path with compile-time K");
        // IN3A: the "fetch" of 4 consecutive channels of row x is the row's three coordinates; this thread's channels are the same
        // in every item it stages (k4 = 4 (tid % 8), THREADS % 8 == 0), so its xyz-layer weights live in registers per chunk
        const auto fa = [&](int x, int k) {
            if constexpr (IN3A) {
                const float *xr = g.x3 + (size_t)(row0 + x) * 3;
                return make_float4(xr[0], xr[1], xr[2], 0.f);
            } else {
                return *reinterpret_cast<const float4 *>(a.z + (size_t)(row0 + x) * Ci + k);
            }
        };
        const auto fb = [&](int x, int k) { return w.template load_ci4<FULL>(col0 + x, k); };
        constexpr int NW3 = IN3A ? KT / BKX : 1;
        float w3r[NW3][4][3], b3r[NW3][4];
        if constexpr (IN3A) {
            const int k4 = (threadIdx.x % (BKX / 4)) * 4;
#pragma unroll
            for (int ch = 0; ch < NW3; ++ch)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int cc = ch * BKX + k4 + j;
                    w3r[ch][j][0] = g.w3[cc * 3], w3r[ch][j][1] = g.w3[cc * 3 + 1], w3r[ch][j][2] = g.w3[cc * 3 + 2];
                    b3r[ch][j] = g.b3 ? g.b3[cc] : 0.f;
                }
        }
#if SN_BF16X3
        constexpr int NCHK = KT > 0 ? KT / BKX : 1;                        // chunks of the whole K (when known)
        constexpr int NCH = PLANES ? (NCHK < 2 ? NCHK : 2) : NCHK;         // chunks fetched up front
        float4 ra[PLANES ? 2 : NCH][Bx3<T>::A4], rb[PLANES ? 1 : NCH][Bx3<T>::B4];
        bf16x8 rp[PLANES ? 2 : 1][Bx3P<T>::NB][3];
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            if constexpr (PLANES) {
                fetch_a_x<T>(ra[ch], fa, ch * BKX, threadIdx.x);
                fetch_planes_x<T>(rp[ch], g.wplanes, Co, Ci, col0, ch * BKX, threadIdx.x);
            } else {
                fetch_chunk_x<T>(ra[ch], rb[ch], fa, fb, ch * BKX, threadIdx.x);
            }
        }
#else
        float4 ra[T::A4], rb[T::B4];
        fetch_chunk<T, true, true>(ra, rb, fa, fb, 0, threadIdx.x);
#endif
        if (c < Ci) {
            long long sa = 0, sb = 0;
#pragma unroll
            for (int q = 0; q < kFxSlots; ++q) sa += lo0[q], sb += lo1[q];
            const double x = (double)sa + (double)ha * kFx2p50, y = (double)sb + (double)hb * kFx2p50;
            const double scl = (1.0 / (double)(1ull << 30)) * (1.0 / (double)(1ull << (kFxShiftFwd - 30)));
            const double nan = __longlong_as_double(0x7ff8000000000000ll);
            const double sum1 = poison ? nan : x * scl, sum2 = poison ? nan : y * scl;
            const double rR = fast_rcp((double)bp.R);
            const double mean = sum1 * rR;
            double var = sum2 * rR - mean * mean;
            if (var < 0.0) var = 0.0;
            const float invstd = (float)fast_rsqrt(var + (double)bp.eps);
            const float sc = bg * invstd, sh = bb - (float)mean * sc;
            cf[c] = sc, cf[Ci + c] = sh;
            if (first) {
                bp.coef[c] = sc, bp.coef[Ci + c] = sh, bp.coef[2 * Ci + c] = (float)mean, bp.coef[3 * Ci + c] = invstd;
                if (bp.running_mean) {
                    const double unbiased = bp.R > 1 ? var * (double)bp.R * fast_rcp((double)(bp.R - 1)) : var;
                    bp.running_mean[c] = (1.f - bp.momentum) * brm + bp.momentum * (float)mean;
                    bp.running_var[c] = (1.f - bp.momentum) * brv + bp.momentum * (float)unbiased;
                }
            }
        }
        if (first && threadIdx.x == 0 && bp.num_batches_tracked) *bp.num_batches_tracked += 1;
        fx_clear_share(g.zero_ptr, g.zero_n, blockIdx.x + gridDim.x * blockIdx.y, gridDim.x * gridDim.y, threadIdx.x, T::THREADS);
        __syncthreads();
        SN_TL(1);
        const auto xa = [&](float4 v, int k) {
            const float4 sc = *reinterpret_cast<const float4 *>(cf + k), sh = *reinterpret_cast<const float4 *>(cf + Ci + k);
            if constexpr (IN3A) {  // v = (x, y, z, -) of the row: the xyz layer's expression, bit for bit (conv_in3_fwd_kernel)
#pragma clang fp contract(off)
                const int ch = k / BKX;  // (a constant after unrolling)
                const float x0 = v.x, x1 = v.y, x2 = v.z;
                v.x = fmaf(w3r[ch][0][2], x2, fmaf(w3r[ch][0][1], x1, w3r[ch][0][0] * x0)) + b3r[ch][0];
                v.y = fmaf(w3r[ch][1][2], x2, fmaf(w3r[ch][1][1], x1, w3r[ch][1][0] * x0)) + b3r[ch][1];
                v.z = fmaf(w3r[ch][2][2], x2, fmaf(w3r[ch][2][1], x1, w3r[ch][2][0] * x0)) + b3r[ch][2];
                v.w = fmaf(w3r[ch][3][2], x2, fmaf(w3r[ch][3][1], x1, w3r[ch][3][0] * x0)) + b3r[ch][3];
            }
            v.x = relu_np(fmaf(v.x, sc.x, sh.x)), v.y = relu_np(fmaf(v.y, sc.y, sh.y));
            v.z = relu_np(fmaf(v.z, sc.z, sh.z)), v.w = relu_np(fmaf(v.w, sc.w, sh.w));
            return v;
        };
#if SN_BF16X3
        if constexpr (PLANES) {
            if constexpr (KT > 0)
                gemm_tile_bx3_ring_p<T, NCHK>(acc, fa, xa, g.wplanes, Co, col0, ra, rp, lds);
            else
                gemm_tile_bx3_p<T>(acc, Ci, fa, xa, g.wplanes, Co, col0, ra[0], rp[0], lds);
        } else if constexpr (KT > 0) {
            gemm_tile_bx3_all<T, NCH>(acc, xa, ra, rb, lds);
        } else {
            gemm_tile_bx3<T>(acc, Ci, fa, fb, xa, ra[0], rb[0], lds);
        }
#else
        gemm_tile_x<T>(acc, Ci, fa, fb, xa, ra, rb, lds);
#endif
    } else {
        const auto fa = [&](int x, int k) { return a.template load_c4<FULL, AMODE == ACT_BN_RELU_FX ? ACT_BN_RELU : AMODE>(row0 + x, k); };
        const auto fb = [&](int x, int k) { return w.template load_ci4<FULL>(col0 + x, k); };
#if SN_BF16X3
        if (FULL && Ci % BKX == 0) {  // same arithmetic as the statistics-chain path above
            float4 ra[Bx3<T>::A4], rb[Bx3<T>::B4];
            fetch_chunk_x<T>(ra, rb, fa, fb, 0, threadIdx.x);
            gemm_tile_bx3<T>(acc, Ci, fa, fb, [](float4 v, int) { return v; }, ra, rb, lds);
        } else
#endif
            gemm_tile<T, true, true>(acc, Ci, fa, fb, lds);
    }

    float s0[T::TN], s1[T::TN];
    float pmax[T::TN], pmin[T::TN];
    int imax[T::TN], imin[T::TN];
    const bool pool = FULL && (g.pool_val != nullptr || g.pool_keys != nullptr);
    const bool store_z = g.z != nullptr;  // (sn_linear_forward_maxpool: only the per-cloud maxima leave the kernel)
    // FULL tiles leave as 16-byte stores: each 32 x 32 fragment is transposed through a per-wave LDS scratch (a dword
    // store per fragment element costs ~58 issue cycles per wave-instruction: 16 of them per fragment were issue-bound)
    float *Ts = lds + 2 * T::WR * T::BN + wave * (32 * 36);  // behind column_reduce2's area; staging buffers are dead
    static_assert(2 * T::WR * T::BN + T::WR * T::WC * 32 * 36 <= T::LDS_FLOATS, "transpose scratch must fit the staging LDS");
#pragma unroll
    for (int j = 0; j < T::TN; ++j) {
        const int col = col0 + (wc * T::TN + j) * 32 + (lane & 31);
        const float bias = biasv[j];
        s0[j] = 0.f, s1[j] = 0.f;
        pmax[j] = -INFINITY, pmin[j] = INFINITY, imax[j] = 0, imin[j] = 0;
#pragma unroll
        for (int i = 0; i < T::TM; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = row0 + (wr * T::TM + i) * 32 + frag_row(e, lane);
                const float v = acc[i][j][e] + bias;
                if (FULL || (row < R && col < Co)) {
                    if (FULL) {
                        if (store_z) Ts[frag_row(e, lane) * 36 + (lane & 31)] = v;
                    } else
                        g.z[(size_t)row * Co + col] = v;
                    s0[j] += v;
                    s1[j] += v * v;
                    if (FULL && pool) {  // rows ascend with e inside a lane: strict compares keep the first occurrence
                        if (v > pmax[j]) pmax[j] = v, imax[j] = row;
                        if (v < pmin[j]) pmin[j] = v, imin[j] = row;
                    }
                }
            }
            if (FULL && store_z) {
                float *zt = g.z + (size_t)(row0 + (wr * T::TM + i) * 32) * Co + col0 + (wc * T::TN + j) * 32 + (lane & 7) * 4;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int rt = 8 * q + (lane >> 3);
                    *reinterpret_cast<float4 *>(zt + (size_t)rt * Co) =
                        *reinterpret_cast<const float4 *>(Ts + rt * 36 + (lane & 7) * 4);
                }
            }
        }
    }
    SN_TL(3);
    SN_TL_DRAIN();
    SN_TL(4);
    if (g.acc_out) {
        // (a layer of 256 output channels: columns 128 .. 255 add into the second accumulator block; col0 is uniform)
        const int blk = col0 / kFxRow;
        column_reduce2<T, true>(s0, s1, lds, reinterpret_cast<float *>(g.acc_out + blk * kFxLayer), nullptr, col0 - blk * kFxRow,
                                Co - blk * kFxRow);
    } else if (g.stats) {
        float *st = g.stats + (size_t)blockIdx.x * 2 * Co;
        column_reduce2<T>(s0, s1, lds, st, st + Co, col0, Co);
    }
    if (FULL && pool && g.pool_keys) {
        // per cloud and column: (maximum, first row) and (minimum, first row) of this wave's 32 rows straight into the cloud's
        // keys (the tile lies inside one cloud: npts % 64 == 0)
        const int cloud = row0 / g.pool_npts, cloud0 = cloud * g.pool_npts;
#pragma unroll
        for (int j = 0; j < T::TN; ++j) {
            const float om = __shfl_xor(pmax[j], 32), on = __shfl_xor(pmin[j], 32);
            const int oim = __shfl_xor(imax[j], 32), oin = __shfl_xor(imin[j], 32);
            if (om > pmax[j] || (om == pmax[j] && oim < imax[j])) pmax[j] = om, imax[j] = oim;
            if (on < pmin[j] || (on == pmin[j] && oin < imin[j])) pmin[j] = on, imin[j] = oin;
            if (lane < 32) {
                const int c = col0 + (wc * T::TN + j) * 32 + lane;
                unsigned long long *kk = g.pool_keys + ((size_t)cloud * 2) * Co + c;
                atomicMax(kk, pool_key(pmax[j], imax[j] - cloud0));
                if (!g.pool_max_only) atomicMax(kk + Co, pool_key(-pmin[j], imin[j] - cloud0));
            }
        }
    } else if (FULL && pool) {
        // block maximum / minimum per column with the first row that attains it: halves of a wave, then the row waves
        __syncthreads();
        float *pv = lds;                                               // [WR][2][BN]
        int *pi = reinterpret_cast<int *>(lds + T::WR * 2 * T::BN);    // [WR][2][BN]
#pragma unroll
        for (int j = 0; j < T::TN; ++j) {
            const float om = __shfl_xor(pmax[j], 32), on = __shfl_xor(pmin[j], 32);
            const int oim = __shfl_xor(imax[j], 32), oin = __shfl_xor(imin[j], 32);
            if (om > pmax[j] || (om == pmax[j] && oim < imax[j])) pmax[j] = om, imax[j] = oim;
            if (on < pmin[j] || (on == pmin[j] && oin < imin[j])) pmin[j] = on, imin[j] = oin;
            if (lane < 32) {
                const int c = (wc * T::TN + j) * 32 + lane;
                pv[(wr * 2 + 0) * T::BN + c] = pmax[j], pv[(wr * 2 + 1) * T::BN + c] = pmin[j];
                pi[(wr * 2 + 0) * T::BN + c] = imax[j], pi[(wr * 2 + 1) * T::BN + c] = imin[j];
            }
        }
        __syncthreads();
        if ((int)threadIdx.x < T::BN) {
            const int c = threadIdx.x;
            float vm = pv[c], vn = pv[T::BN + c];
            int im = pi[c], in_ = pi[T::BN + c];
#pragma unroll
            for (int r = 1; r < T::WR; ++r) {  // row waves hold ascending rows: strict compares keep the first occurrence
                const float a = pv[(r * 2 + 0) * T::BN + c], b2 = pv[(r * 2 + 1) * T::BN + c];
                const int ia = pi[(r * 2 + 0) * T::BN + c], ib = pi[(r * 2 + 1) * T::BN + c];
                if (a > vm || (a == vm && ia < im)) vm = a, im = ia;
                if (b2 < vn || (b2 == vn && ib < in_)) vn = b2, in_ = ib;
            }
            const int cloud0 = (row0 / g.pool_npts) * g.pool_npts;
            float *ov = g.pool_val + (size_t)blockIdx.x * 2 * Co + col0 + c;
            int *oi = g.pool_idx + (size_t)blockIdx.x * 2 * Co + col0 + c;
            ov[0] = vm, ov[Co] = vn;
            oi[0] = im - cloud0, oi[Co] = in_ - cloud0;
        }
    }
    SN_TL(5);
}

// ------------------------------------------------------------------------------------------------
// Persistent, weight-stationary form of the statistics-chain forward GEMM for LARGE batches (round 4).  linear_fwd_kernel runs
// one workgroup per 64-row tile; at B = 512 that is 8192 workgroups per layer, each of which pays the statistics prologue (the
// input BatchNorm's fixed-point sums -- lines that were just updated by atomics --, the coefficient arithmetic, the first operand
// round trip: 5 us), re-stages all of W's planes from L2 (98 KB for 32 KB of activations) and ends with its own set of
// statistics / pool atomics (2 us): 15 us of workgroup lifetime for 0.4 us of MFMAs (tools/timeline.py stack 512).  Here ONE
// workgroup per CU (two for the 64-column layers) walks over a contiguous range of tiles:
//   * the input BatchNorm is finalised once; the weights' split planes are read ONCE, straight into the B fragments of the
//     wave's 32 columns, which stay in registers (3 planes x K / 16 fragments: 96 VGPRs at K = 128) -- the MFMA loop reads only
//     A fragments from LDS;
//   * the activations of the next two tiles (all K chunks: one or two 16-byte loads per thread and chunk, two register sets) are
//     in flight under the staging / MFMAs / stores of the current one;
//   * a WHOLE tile's A planes are double-buffered in LDS: tile t + 1 is activated, split and stored while the other waves still
//     multiply tile t -- ONE LDS-only barrier per tile (global loads and stores stay in flight across it), 6 K / 16 MFMAs per
//     wave back to back (per-chunk staging with a barrier per chunk: 154 instead of 200 us for the 128 -> 128 layer at B = 512,
//     a chain of LDS write -> barrier -> read latencies; prefetching two tiles ahead instead of one changed nothing);
//   * column sums are converted to fixed point per tile -- exactly as fx_add does -- and added up in registers; ONE atomic per
//     column and workgroup leaves at the end (integer addition is associative: the totals are those of the per-tile kernel, bit
//     for bit); pool keys are combined in registers over the 16 consecutive tiles of a cloud and published once per cloud.
// Same products in the same order as gemm_tile_bx3: pre-activations, statistics and pooled features are bit-identical to
// linear_fwd_kernel's (tests/test_gpu_mlp.py::test_persistent_forward_is_bit_identical).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void lds_only_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int SHIFT>
__device__ __forceinline__ void fx_local_add(long long &lo, long long *layer, int stat, int c, float v)
{
    const double d = (double)v * (double)(1ull << 30) * (double)(1ull << (SHIFT - 30));
    if (fabs(d) < kFx2p50) {
        lo += __double2ll_rn(d);
    } else if (fabs(d) < kFx2p50 * kFx2p50) {  // (rare: the hi part goes out at once, as in fx_add)
        const double hh = floor(d * (1.0 / kFx2p50));
        lo += __double2ll_rn(d - hh * kFx2p50);
        atomicAdd(reinterpret_cast<unsigned long long *>(layer + kFxHi + stat * kFxRow + c), (unsigned long long)(long long)hh);
    } else {
        layer[kFxPoison] = 1;
    }
}

#ifndef SN_FWDP_ABL
#define SN_FWDP_ABL 0  // (timing experiments only: 1 no MFMAs, 2 no staging of the next tile, 3 no stores of Z, 6 two accumulator chains)
#endif
template <class T, int KT, bool IN3A, int WPE>
__global__ void __launch_bounds__(T::THREADS) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) linear_fwd_persist_kernel(FwdArgs g, int ntiles,
                                                                                                                         int tpw)
{
#if SN_BF16X3
    static_assert(T::TM == 1 && T::TN == 1 && KT % BKX == 0 && KT <= 128, "one 32 x 32 block per wave");
    constexpr int NCH = KT / BKX, A4 = Bx3<T>::A4, KS = KT / 16, Ci = KT;
    constexpr int ACH = 3 * T::BM * LDX;  // bf16 elements of one chunk's three planes
    constexpr int ABUF = NCH * ACH;       // ... of a whole tile (all K chunks)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __bf16 *Abuf = reinterpret_cast<__bf16 *>(lds);  // [2][NCH][3][BM][LDX]: two whole tiles
    float *cf = lds + ABUF;                           // [2][Ci]   (2 buffers x ABUF bf16 = ABUF floats)
    float *red = cf + 2 * Ci;                         // [2][WR][2][BN]  (by tile parity: a fast wave may reach the next tile's
                                                      //  sums while a slow one still reads this tile's)
    float *TsAll = red + 2 * T::WR * 2 * T::BN;       // [waves][16 x 36] transposes of the output fragments, half a fragment at a time
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / T::WC, wc = wave % T::WC;
    const int Co = g.w.co;
    const int tile0 = blockIdx.x * tpw, tile1 = min(ntiles, tile0 + tpw);
    const bool first = blockIdx.x == 0;
    const BnFwd bp = g.bn_prev;
    // ---- prologue: the input BatchNorm's sums, the first tile's activations, the weights' fragments -- all requested before
    // anything is waited for
    const int c = tid;
    long long lo0[kFxSlots], lo1[kFxSlots], ha = 0, hb = 0, poison = 0;
    float bg = 0.f, bb = 0.f, brm = 0.f, brv = 0.f;
    if (c < Ci) {
        poison = g.acc_in[kFxPoison];
#pragma unroll
        for (int q = 0; q < kFxSlots; ++q) lo0[q] = g.acc_in[(q * 2 + 0) * kFxRow + c], lo1[q] = g.acc_in[(q * 2 + 1) * kFxRow + c];
        ha = g.acc_in[kFxHi + c], hb = g.acc_in[kFxHi + kFxRow + c];
        bg = bp.gamma[c], bb = bp.beta[c];
        if (first && bp.running_mean) brm = bp.running_mean[c], brv = bp.running_var[c];
    }
    const int colw = wc * 32 + l31;  // this lane's output column (T::BN == Co)
    const float biasv = g.bias ? g.bias[colw] : 0.f;
    float w3r[IN3A ? NCH : 1][4][3], b3r[IN3A ? NCH : 1][4];
    if constexpr (IN3A) {
        const int k4 = (tid % (BKX / 4)) * 4;
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int cc = ch * BKX + k4 + j;
                w3r[ch][j][0] = g.w3[cc * 3], w3r[ch][j][1] = g.w3[cc * 3 + 1], w3r[ch][j][2] = g.w3[cc * 3 + 2];
                b3r[ch][j] = g.b3 ? g.b3[cc] : 0.f;
            }
    }
    const auto fetch_tile = [&](float4 (&ra)[NCH][A4], int tile) {
        const size_t r0 = (size_t)tile * T::BM;
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
            for (int q = 0; q < A4; ++q) {
                const int f = tid + q * T::THREADS, x = bx3_row(f), k = ch * BKX + (f % (BKX / 4)) * 4;
                if constexpr (IN3A) {
                    const float *xr = g.x3 + (r0 + x) * 3;
                    ra[ch][q] = make_float4(xr[0], xr[1], xr[2], 0.f);
                } else {
                    ra[ch][q] = *reinterpret_cast<const float4 *>(g.a.z + (r0 + x) * Ci + k);
                }
            }
    };
    // two register sets: tiles t + 1 and t + 2 (later t + 2 and t + 3) are in flight while tile t is multiplied
    float4 ra0[NCH][A4], ra1[NCH][A4];
    if (tile0 < tile1) fetch_tile(ra0, tile0);
    if (tile0 + 1 < tile1) fetch_tile(ra1, tile0 + 1);
    bf16x8 breg[KS][3];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int p = 0; p < 3; ++p)
            breg[ks][p] = *reinterpret_cast<const bf16x8 *>(g.wplanes + ((size_t)p * Co + colw) * Ci + ks * 16 + 8 * h);
    if (c < Ci) {  // (same arithmetic as linear_fwd_kernel's prologue)
        long long sa = 0, sb = 0;
#pragma unroll
        for (int q = 0; q < kFxSlots; ++q) sa += lo0[q], sb += lo1[q];
        const double x = (double)sa + (double)ha * kFx2p50, y = (double)sb + (double)hb * kFx2p50;
        const double scl = (1.0 / (double)(1ull << 30)) * (1.0 / (double)(1ull << (kFxShiftFwd - 30)));
        const double nan = __longlong_as_double(0x7ff8000000000000ll);
        const double sum1 = poison ? nan : x * scl, sum2 = poison ? nan : y * scl;
        const double rR = fast_rcp((double)bp.R);
        const double mean = sum1 * rR;
        double var = sum2 * rR - mean * mean;
        if (var < 0.0) var = 0.0;
        const float invstd = (float)fast_rsqrt(var + (double)bp.eps);
        const float sc = bg * invstd, sh = bb - (float)mean * sc;
        cf[c] = sc, cf[Ci + c] = sh;
        if (first) {
            bp.coef[c] = sc, bp.coef[Ci + c] = sh, bp.coef[2 * Ci + c] = (float)mean, bp.coef[3 * Ci + c] = invstd;
            if (bp.running_mean) {
                const double unbiased = bp.R > 1 ? var * (double)bp.R * fast_rcp((double)(bp.R - 1)) : var;
                bp.running_mean[c] = (1.f - bp.momentum) * brm + bp.momentum * (float)mean;
                bp.running_var[c] = (1.f - bp.momentum) * brv + bp.momentum * (float)unbiased;
            }
        }
    }
    if (first && tid == 0 && bp.num_batches_tracked) *bp.num_batches_tracked += 1;
    fx_clear_share(g.zero_ptr, g.zero_n, blockIdx.x, gridDim.x, tid, T::THREADS);
    __syncthreads();
    const auto xa = [&](float4 v, int k) {
        const float4 sc = *reinterpret_cast<const float4 *>(cf + k), sh = *reinterpret_cast<const float4 *>(cf + Ci + k);
        if constexpr (IN3A) {  // v = (x, y, z, -) of the row: the xyz layer's expression, bit for bit (conv_in3_fwd_kernel)
#pragma clang fp contract(off)
            const int ch = k / BKX;
            const float x0 = v.x, x1 = v.y, x2 = v.z;
            v.x = fmaf(w3r[ch][0][2], x2, fmaf(w3r[ch][0][1], x1, w3r[ch][0][0] * x0)) + b3r[ch][0];
            v.y = fmaf(w3r[ch][1][2], x2, fmaf(w3r[ch][1][1], x1, w3r[ch][1][0] * x0)) + b3r[ch][1];
            v.z = fmaf(w3r[ch][2][2], x2, fmaf(w3r[ch][2][1], x1, w3r[ch][2][0] * x0)) + b3r[ch][2];
            v.w = fmaf(w3r[ch][3][2], x2, fmaf(w3r[ch][3][1], x1, w3r[ch][3][0] * x0)) + b3r[ch][3];
        }
        v.x = relu_np(fmaf(v.x, sc.x, sh.x)), v.y = relu_np(fmaf(v.y, sc.y, sh.y));
        v.z = relu_np(fmaf(v.z, sc.z, sh.z)), v.w = relu_np(fmaf(v.w, sc.w, sh.w));
        return v;
    };
    // a whole tile: activated, split and stored into buffer `buf` (all K chunks, chunk-major as bx3_chunk_g lays one out)
    const auto stage = [&](const float4 (&ra)[NCH][A4], int buf) {
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            __bf16 *Ap = Abuf + buf * ABUF + ch * ACH;
#pragma unroll
            for (int q = 0; q < A4; ++q) {
                const int f = tid + q * T::THREADS, k4 = (f % (BKX / 4)) * 4;
                stage_split<T::BM>(Ap, bx3_row(f), k4, xa(ra[ch][q], ch * BKX + k4));
            }
        }
    };
    // running per-column sums (threads tid < BN) and the current cloud's pool keys (lanes < 32 of every wave)
    // (HT, 32-row tiles: the two halves of a 64-row block are consecutive tiles of one workgroup; the first half's column sums wait
    //  in pend0 / pend1 and the pair is converted to fixed point as ONE 64-row sum -- (rows 0..31) + (rows 32..63), the per-tile
    //  kernel's own order: the totals stay bit-identical)
    constexpr bool HT = T::BM == 32;
    float pend0 = 0.f, pend1 = 0.f;
    long long fx0 = 0, fx1 = 0;
    unsigned long long kmx = 0ull, kmn = 0ull;
    int kcloud = -1;
    const bool pool = g.pool_keys != nullptr;
    float *Ts = TsAll + wave * (16 * 36);
    const auto flush_keys = [&]() {
        if (pool && kcloud >= 0 && lane < 32) {
            unsigned long long *kk = g.pool_keys + ((size_t)kcloud * 2) * Co + colw;
            atomicMax(kk, kmx);
            if (!g.pool_max_only) atomicMax(kk + Co, kmn);
        }
    };
    // Iteration t: tile t + 1 is activated / split / stored into the OTHER buffer (its loads were issued two iterations ago), its
    // register set is refilled with tile t + 3, then tile t is multiplied out of its buffer -- 6 K / 16 MFMAs per wave back to
    // back, only A fragments read from LDS -- and leaves through the epilogue; ONE barrier per tile: the staging of a wave
    // overlaps the MFMAs of the others.
    const auto process = [&](int tile, float4 (&nxt)[NCH][A4]) {
        const int buf = (tile - tile0) & 1;
        if (tile + 1 < tile1 && SN_FWDP_ABL != 2) stage(nxt, buf ^ 1);
        if (tile + 3 < tile1) fetch_tile(nxt, tile + 3);
        const int row0 = tile * T::BM;
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
        const __bf16 *At = Abuf + buf * ABUF;
#if SN_FWDP_ABL == 6
        f32x16 acc2;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc2[e] = 0.f;
#endif
#pragma unroll
        for (int ks = 0; ks < (SN_FWDP_ABL == 1 ? 0 : KS); ++ks) {
            const __bf16 *Ap = At + (ks / (BKX / 16)) * ACH;
            const int kk = ks % (BKX / 16);
            bf16x8 a[3];
#pragma unroll
            for (int p = 0; p < 3; ++p) a[p] = *reinterpret_cast<const bf16x8 *>(Ap + (p * T::BM + wr * 32 + l31) * LDX + kk * 16 + 8 * h);
            // smallest products first (the order of bx3_chunk_g)
#if SN_FWDP_ABL == 6
            if (ks & 1) {
                acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], breg[ks][2], acc2, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], breg[ks][0], acc2, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], breg[ks][1], acc2, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], breg[ks][1], acc2, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], breg[ks][0], acc2, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], breg[ks][0], acc2, 0, 0, 0);
                continue;
            }
#endif
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], breg[ks][2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], breg[ks][0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], breg[ks][1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], breg[ks][1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], breg[ks][0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], breg[ks][0], acc, 0, 0, 0);
        }
#if SN_FWDP_ABL == 6
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] += acc2[e];
#endif
        // ---- epilogue: bias, column sums, pool candidates, 16-byte stores through the wave's transpose scratch
        float s0 = 0.f, s1 = 0.f, pmax = -INFINITY, pmin = INFINITY;
        int imax = 0, imin = 0;
        float *redt = red + ((tile - tile0) & 1) * (T::WR * 2 * T::BN);
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {  // rows 16 hf .. 16 hf + 15 of the fragment: accumulator registers 8 hf .. 8 hf + 7
#pragma unroll
            for (int e8 = 0; e8 < 8; ++e8) {
                const int e = hf * 8 + e8;
                const int row = row0 + wr * 32 + frag_row(e, lane);
                const float v = acc[e] + biasv;
                Ts[(frag_row(e, lane) - 16 * hf) * 36 + l31] = v;
                s0 += v;
                s1 += v * v;
                if (v > pmax) pmax = v, imax = row;
                if (v < pmin) pmin = v, imin = row;
            }
            if (g.z && SN_FWDP_ABL != 3) {
                float *zt = g.z + (size_t)(row0 + wr * 32 + 16 * hf) * Co + wc * 32 + (lane & 7) * 4;
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int rt = 8 * q + (lane >> 3);
                    *reinterpret_cast<float4 *>(zt + (size_t)rt * Co) = *reinterpret_cast<const float4 *>(Ts + rt * 36 + (lane & 7) * 4);
                }
            }
        }
        {  // column_reduce2: halves of a wave, then the row waves in index order
            const float t0 = s0 + __shfl_xor(s0, 32), t1 = s1 + __shfl_xor(s1, 32);
            if (lane < 32) redt[(wr * 2 + 0) * T::BN + colw] = t0, redt[(wr * 2 + 1) * T::BN + colw] = t1;
        }
        if (pool) {
            const int cloud = row0 / g.pool_npts, cloud0 = cloud * g.pool_npts;
            const float om = __shfl_xor(pmax, 32), on = __shfl_xor(pmin, 32);
            const int oim = __shfl_xor(imax, 32), oin = __shfl_xor(imin, 32);
            if (om > pmax || (om == pmax && oim < imax)) pmax = om, imax = oim;
            if (on < pmin || (on == pmin && oin < imin)) pmin = on, imin = oin;
            if (cloud != kcloud) {
                flush_keys();
                kcloud = cloud, kmx = 0ull, kmn = 0ull;
            }
            const unsigned long long k1 = pool_key(pmax, imax - cloud0), k2 = pool_key(-pmin, imin - cloud0);
            kmx = k1 > kmx ? k1 : kmx, kmn = k2 > kmn ? k2 : kmn;
        }
        lds_only_barrier();
        if (tid < T::BN) {
            float a0 = 0.f, a1 = 0.f;
#pragma unroll
            for (int r = 0; r < T::WR; ++r) a0 += redt[(r * 2 + 0) * T::BN + tid], a1 += redt[(r * 2 + 1) * T::BN + tid];
            if (HT && ((tile - tile0) & 1) == 0) {
                pend0 = a0, pend1 = a1;
            } else {
                if (HT) a0 = pend0 + a0, a1 = pend1 + a1;
                fx_local_add<kFxShiftFwd>(fx0, g.acc_out, 0, tid, a0);
                fx_local_add<kFxShiftFwd>(fx1, g.acc_out, 1, tid, a1);
            }
        }
    };
    if (tile0 < tile1) {  // tile0 into buffer 0; its register set takes tile0 + 2
        stage(ra0, 0);
        if (tile0 + 2 < tile1) fetch_tile(ra0, tile0 + 2);
        lds_only_barrier();
    }
    for (int tile = tile0; tile < tile1; tile += 2) {  // (tile t + 1 waits in set (t + 1 - tile0) & 1)
        process(tile, ra1);
        if (tile + 1 < tile1) process(tile + 1, ra0);
    }
    flush_keys();
    if (tid < T::BN && tile0 < tile1) {
        const int slot = blockIdx.x % kFxSlots;
        atomicAdd(reinterpret_cast<unsigned long long *>(g.acc_out + (slot * 2 + 0) * kFxRow + tid), (unsigned long long)fx0);
        atomicAdd(reinterpret_cast<unsigned long long *>(g.acc_out + (slot * 2 + 1) * kFxRow + tid), (unsigned long long)fx1);
    }
#endif
}

// ------------------------------------------------------------------------------------------------
// Small-R kernels (R <= 32: the FC head at the reference batch size).  One 32-row MFMA tile; the GEMM is
// latency-bound, so there is no LDS staging loop: every lane loads its whole K slice of both operands straight
// into registers (all loads in flight at once), the four waves of a workgroup split K, and their accumulators
// are summed through LDS in wave order (deterministic).  MFMA step t of a lane consumes k = kbase + 32*half + t:
// any permutation of k is valid as long as A and B use the same one.
// ------------------------------------------------------------------------------------------------

// Z[R<=32][Co] = act(A) . W^T + bias ; stats [1][2][Co]
// VEC: Ci % 64 == 0 -> every executed pass is fully in range and 16-byte aligned: the lane's 32 consecutive k of a row
// are fetched as 8 dwordx4 loads (one 128-byte line per row, touched once) instead of 32 strided dword loads.
template <int AMODE, bool VEC>
__global__ void __launch_bounds__(256) small_fwd_kernel(FwdArgs g)
{
    __shared__ float lds[3 * 16 * 64];
    SN_TL(0);
    const int lane = threadIdx.x & 63, l31 = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int R = g.a.rows, Co = g.w.co, Ci = g.w.ci;
    const int col = blockIdx.x * 32 + l31;
    const bool colok = col < Co;
    const int ccol = colok ? col : 0;
    const float *wrow = g.w.w + (size_t)ccol * Ci;
    // epilogue inputs first: loads issued after the MFMAs would each expose a full memory latency
    const float bias = g.bias ? g.bias[ccol] : 0.f;
    float bn_g = 0.f, bn_b = 0.f, bn_rm = 0.f, bn_rv = 0.f;
    if (g.bn.coef) {
        bn_g = g.bn.gamma[ccol], bn_b = g.bn.beta[ccol];
        if (g.bn.running_mean) bn_rm = g.bn.running_mean[ccol], bn_rv = g.bn.running_var[ccol];
    }
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    for (int k0 = wave * 2 * KP; k0 < Ci; k0 += 4 * 2 * KP) {
        float a[KP], b[KP];
        const int kb = k0 + h * KP;
        if (VEC) {
            const int rr = l31 < R ? l31 : 0;
            const float rmask = l31 < R ? 1.f : 0.f, cmask = colok ? 1.f : 0.f;
#pragma unroll
            for (int t = 0; t < KP; t += 4) {
                const float4 av = g.a.template load_c4<true, AMODE>(rr, kb + t);
                const float4 bv = *reinterpret_cast<const float4 *>(wrow + kb + t);
                a[t] = av.x * rmask, a[t + 1] = av.y * rmask, a[t + 2] = av.z * rmask, a[t + 3] = av.w * rmask;
                b[t] = bv.x * cmask, b[t + 1] = bv.y * cmask, b[t + 2] = bv.z * cmask, b[t + 3] = bv.w * cmask;
            }
        } else {
#pragma unroll
            for (int t = 0; t < KP; ++t) {
                const int k = kb + t;
                a[t] = g.a.template at<AMODE>(l31, k);
                // multiply by a 0/1 mask instead of selecting: a select lets the compiler sink the load into a
                // branch and wait for it on the spot, serialising all 32 loads
                b[t] = wrow[k < Ci ? k : 0] * ((colok && k < Ci) ? 1.f : 0.f);
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
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int row = frag_row(e, lane);
        const float v = acc[e] + bias;
        if (row < R && colok) {
            g.z[(size_t)row * Co + col] = v;
            s0 += v;
            s1 += v * v;
        }
    }
    s0 += __shfl_xor(s0, 32);
    s1 += __shfl_xor(s1, 32);
    if (g.stats && lane < 32 && colok) g.stats[col] = s0, g.stats[Co + col] = s1;
    if (g.bn.coef) {  // this workgroup holds every row of its 32 columns: their batch statistics are complete here
        if (blockIdx.x == 0 && lane == 0 && g.bn.num_batches_tracked) *g.bn.num_batches_tracked += 1;
        // Two-pass variance: the rows are all in this wave's registers, so the squares are summed around the mean.  (The FC head
        // sits behind the max-pool: its pre-BN features are nearly the same for every cloud of a batch -- |mean| / std of 10..100 --
        // and E[z^2] - mean^2 from fp32 sums then loses 2..4 digits of the variance; measured 5x the error of torch's CPU path
        // at the head's output before this.)
        const float meanf = s0 / (float)g.bn.R;
        float s2 = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const float d = (acc[e] + bias) - meanf;
            if (frag_row(e, lane) < R && colok) s2 += d * d;
        }
        s2 += __shfl_xor(s2, 32);
        if (lane < 32 && colok) {  // same arithmetic as bn_finalize_channel, on the prefetched parameters
            const double rR = fast_rcp((double)g.bn.R);
            const double mean = (double)s0 * rR;
            const double dm = mean - (double)meanf;  // sum (z - meanf)^2 = sum (z - mean)^2 + R dm^2
            double var = (double)s2 * rR - dm * dm;
            if (var < 0.0) var = 0.0;
            const float invstd = (float)fast_rsqrt(var + (double)g.bn.eps);
            const float sc = bn_g * invstd;
            g.bn.coef[col] = sc;
            g.bn.coef[Co + col] = bn_b - (float)mean * sc;
            g.bn.coef[2 * Co + col] = (float)mean;
            g.bn.coef[3 * Co + col] = invstd;
            if (g.bn.running_mean) {
                const double unbiased = g.bn.R > 1 ? var * (double)g.bn.R * fast_rcp((double)(g.bn.R - 1)) : var;
                g.bn.running_mean[col] = (1.f - g.bn.momentum) * bn_rm + g.bn.momentum * (float)mean;
                g.bn.running_var[col] = (1.f - g.bn.momentum) * bn_rv + g.bn.momentum * (float)unbiased;
            }
        }
    }
    SN_TL_DRAIN();
    SN_TL(5);
}

// dYprev[R<=32][Ci] = mask . (dZ . W) ; stats [1][2][Ci]
// small_fwd_kernel with both operands staged through LDS (Ci % 64 == 0, Ci <= 512).  In the register-direct version every
// lane fetches its own 128-byte stretch of a row: 64 cache lines per wave-instruction, 16 such instructions per wave --
// measured 3.4 us from launch to "operands landed" for a 32 x 256 x 256 layer.  Here the 256 threads fetch the two 32-row
// slabs with fully coalesced 16-byte loads (BatchNorm + ReLU of the previous layer applied on the way), fragments come
// from LDS as ds_read_b128 (row pitch Ci + 4: conflict-free), and the output tile leaves as 16-byte stores.
template <int AMODE>
__global__ void __launch_bounds__(256) small_fwd_lds_kernel(FwdArgs g)
{
    extern __shared__ __attribute__((aligned(16))) float sm[];
    SN_TL(0);
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int R = g.a.rows, Co = g.w.co, Ci = g.w.ci;
    const int LD = Ci + 4;
    float *As = sm, *Ws = sm + 32 * LD, *red = Ws + 32 * LD, *Ts = red + 3 * 16 * 64;
    const int col0 = blockIdx.x * 32;
    // gridDim.y > 1 (32 < R, launch_fwd's row-tiled case): workgroup (x, y) owns rows [32 y, 32 y + 32) -- the statistics of a
    // column are then spread over gridDim.y workgroups, so neither stats nor the BatchNorm finalisation may be requested
    const int rb = blockIdx.y * 32;
    const int col = col0 + l31;
    const bool colok = col < Co;
    const int ccol = colok ? col : 0;
    // epilogue inputs first: loads issued after the MFMAs would each expose a full memory latency
    const float bias = g.bias ? g.bias[ccol] : 0.f;
    float bn_g = 0.f, bn_b = 0.f, bn_rm = 0.f, bn_rv = 0.f;
    if (g.bn.coef) {
        bn_g = g.bn.gamma[ccol], bn_b = g.bn.beta[ccol];
        if (g.bn.running_mean) bn_rm = g.bn.running_mean[ccol], bn_rv = g.bn.running_var[ccol];
    }
    // ---- stage: thread -> (row = tid / (Ci/4) + q * rows_per_pass, 4 channels at c4), the same c4 for every q
    const int q4 = Ci / 4, rpp = 256 / q4, npass = 32 / rpp;  // Ci = 256: 64, 4, 8
    const int c4 = (tid % q4) * 4, r0 = tid / q4;
    float4 sc4 = make_float4(1.f, 1.f, 1.f, 1.f), sh4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (AMODE == ACT_BN_RELU) {
        sc4 = *reinterpret_cast<const float4 *>(g.a.scale + c4);
        sh4 = *reinterpret_cast<const float4 *>(g.a.shift + c4);
    }
    constexpr int MAXP = 16;  // Ci >= 64 -> at most 16 passes per operand
    float4 av[MAXP], wv[MAXP];
#pragma unroll
    for (int q = 0; q < MAXP; ++q)
        if (q < npass) {
            const int r = r0 + q * rpp;
            av[q] = *reinterpret_cast<const float4 *>(g.a.z + (size_t)min(rb + r, R - 1) * Ci + c4);
            wv[q] = *reinterpret_cast<const float4 *>(g.w.w + (size_t)min(col0 + r, Co - 1) * Ci + c4);
        }
#pragma unroll
    for (int q = 0; q < MAXP; ++q)
        if (q < npass) {
            const int r = r0 + q * rpp;
            float4 a = av[q];
            if (AMODE == ACT_BN_RELU) {
                a.x = relu_np(fmaf(a.x, sc4.x, sh4.x)), a.y = relu_np(fmaf(a.y, sc4.y, sh4.y));
                a.z = relu_np(fmaf(a.z, sc4.z, sh4.z)), a.w = relu_np(fmaf(a.w, sc4.w, sh4.w));
            }
            const float ma = rb + r < R ? 1.f : 0.f, mw = col0 + r < Co ? 1.f : 0.f;
            a.x *= ma, a.y *= ma, a.z *= ma, a.w *= ma;
            float4 w = wv[q];
            w.x *= mw, w.y *= mw, w.z *= mw, w.w *= mw;
            *reinterpret_cast<float4 *>(As + r * LD + c4) = a;
            *reinterpret_cast<float4 *>(Ws + r * LD + c4) = w;
        }
    __syncthreads();
    SN_TL(1);
    // ---- MFMA: the four waves split K; lane (l31, h) walks k = kb .. kb + kph - 1 of row / column l31
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    const int kph = Ci / 8, kb = wave * (Ci / 4) + h * kph;
    const float *ap = As + l31 * LD + kb, *bp = Ws + l31 * LD + kb;

    for (int t = 0; t < kph; t += 4) {
        const float4 a = *reinterpret_cast<const float4 *>(ap + t), b = *reinterpret_cast<const float4 *>(bp + t);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
    }
    SN_TL(2);
    wave_sum_to_wave0(acc, red);
    SN_TL(3);
    if (wave != 0) return;
    float s0 = 0.f, s1 = 0.f;
    const bool vec_out = Co % 32 == 0;  // whole 32-column blocks, 16-byte aligned rows
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int row = frag_row(e, lane);
        const float v = acc[e] + bias;
        if (rb + row < R && colok) {
            s0 += v;
            s1 += v * v;
            if (!vec_out) g.z[(size_t)(rb + row) * Co + col] = v;
        }
        Ts[row * 36 + l31] = v;
    }
    if (vec_out) {  // transposed through LDS: 4 x 16-byte stores per lane instead of 16 dword stores
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = 8 * i + (lane >> 3);
            const float4 v = *reinterpret_cast<const float4 *>(Ts + row * 36 + (lane & 7) * 4);
            if (rb + row < R) *reinterpret_cast<float4 *>(g.z + (size_t)(rb + row) * Co + col0 + (lane & 7) * 4) = v;
        }
    }
    s0 += __shfl_xor(s0, 32);
    s1 += __shfl_xor(s1, 32);
    if (g.stats && lane < 32 && colok) g.stats[col] = s0, g.stats[Co + col] = s1;
    if (g.bn.coef) {  // this workgroup holds every row of its 32 columns: their batch statistics are complete here
        if (blockIdx.x == 0 && lane == 0 && g.bn.num_batches_tracked) *g.bn.num_batches_tracked += 1;
        // Two-pass variance: the rows are all in this wave's registers, so the squares are summed around the mean.  (The FC head
        // sits behind the max-pool: its pre-BN features are nearly the same for every cloud of a batch -- |mean| / std of 10..100 --
        // and E[z^2] - mean^2 from fp32 sums then loses 2..4 digits of the variance; measured 5x the error of torch's CPU path
        // at the head's output before this.)
        const float meanf = s0 / (float)g.bn.R;
        float s2 = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const float d = (acc[e] + bias) - meanf;
            if (frag_row(e, lane) < R && colok) s2 += d * d;
        }
        s2 += __shfl_xor(s2, 32);
        // same arithmetic as bn_finalize_channel, on the prefetched parameters (both lanes of a column evaluate it: each holds
        // 16 of the column's rows for the normalised output below; one of them stores)
        const double rR = fast_rcp((double)g.bn.R);
        const double mean = (double)s0 * rR;
        const double dm = mean - (double)meanf;  // sum (z - meanf)^2 = sum (z - mean)^2 + R dm^2
        double var = (double)s2 * rR - dm * dm;
        if (var < 0.0) var = 0.0;
        const float invstd = (float)fast_rsqrt(var + (double)g.bn.eps);
        const float sc = bn_g * invstd, sh = bn_b - (float)mean * sc;
        if (lane < 32 && colok) {
            g.bn.coef[col] = sc;
            g.bn.coef[Co + col] = sh;
            g.bn.coef[2 * Co + col] = (float)mean;
            g.bn.coef[3 * Co + col] = invstd;
            if (g.bn.running_mean) {
                const double unbiased = g.bn.R > 1 ? var * (double)g.bn.R * fast_rcp((double)(g.bn.R - 1)) : var;
                g.bn.running_mean[col] = (1.f - g.bn.momentum) * bn_rm + g.bn.momentum * (float)mean;
                g.bn.running_var[col] = (1.f - g.bn.momentum) * bn_rv + g.bn.momentum * (float)unbiased;
            }
        }
        if (g.bn_y) {  // y = z scale + shift (no activation) through the same transposing tile as z
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = frag_row(e, lane);
                const float y = fmaf(acc[e] + bias, sc, sh);
                if (!vec_out && rb + row < R && colok) g.bn_y[(size_t)(rb + row) * Co + col] = y;
                Ts[row * 36 + l31] = y;
            }
            if (vec_out) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int row = 8 * i + (lane >> 3);
                    const float4 v = *reinterpret_cast<const float4 *>(Ts + row * 36 + (lane & 7) * 4);
                    if (rb + row < R) *reinterpret_cast<float4 *>(g.bn_y + (size_t)(rb + row) * Co + col0 + (lane & 7) * 4) = v;
                }
            }
        }
    }
    SN_TL_DRAIN();
    SN_TL(5);
}

// 33 .. 64 rows (the FC head just above the chain's 32 clouds): small_fwd_lds_kernel with BOTH 32-row halves in one workgroup --
// the weight slice is staged once and feeds two accumulators, and the workgroup holds every row of its 32 columns, so the
// BatchNorm finalisation (two-pass variance in registers, as there) stays in the epilogue: no statistics launch behind the GEMM
// (4.9 us per layer; the step costs 0.19 ms at 32 clouds and this range is where a batch of 48 or 64 lands).  Ci = 64, 128 or 256
// (Ci / 4 threads stage a row: must divide 256).
template <int AMODE>
__global__ void __launch_bounds__(256) rows64_fwd_kernel(FwdArgs g)
{
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int R = g.a.rows, Co = g.w.co, Ci = g.w.ci;
    const int LD = Ci + 4;
    float *As = sm, *Ws = sm + 64 * LD, *red = Ws + 32 * LD, *Ts = red + 2 * 3 * 16 * 64;
    const int col0 = blockIdx.x * 32;
    const int col = col0 + l31;
    const bool colok = col < Co;
    const int ccol = colok ? col : 0;
    const float bias = g.bias ? g.bias[ccol] : 0.f;
    float bn_g = 0.f, bn_b = 0.f, bn_rm = 0.f, bn_rv = 0.f;
    if (g.bn.coef) {
        bn_g = g.bn.gamma[ccol], bn_b = g.bn.beta[ccol];
        if (g.bn.running_mean) bn_rm = g.bn.running_mean[ccol], bn_rv = g.bn.running_var[ccol];
    }
    const int q4 = Ci / 4, rpp = 256 / q4, npa = 64 / rpp, npw = 32 / rpp;  // Ci = 256: 64, 4, 16, 8
    const int c4 = (tid % q4) * 4, r0 = tid / q4;
    float4 sc4 = make_float4(1.f, 1.f, 1.f, 1.f), sh4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (AMODE == ACT_BN_RELU) {
        sc4 = *reinterpret_cast<const float4 *>(g.a.scale + c4);
        sh4 = *reinterpret_cast<const float4 *>(g.a.shift + c4);
    }
    constexpr int MAXA = 16, MAXW = 8;  // Ci >= 64
    float4 av[MAXA], wv[MAXW];
#pragma unroll
    for (int q = 0; q < MAXA; ++q)
        if (q < npa) av[q] = *reinterpret_cast<const float4 *>(g.a.z + (size_t)min(r0 + q * rpp, R - 1) * Ci + c4);
#pragma unroll
    for (int q = 0; q < MAXW; ++q)
        if (q < npw) wv[q] = *reinterpret_cast<const float4 *>(g.w.w + (size_t)min(col0 + r0 + q * rpp, Co - 1) * Ci + c4);
#pragma unroll
    for (int q = 0; q < MAXA; ++q)
        if (q < npa) {
            const int r = r0 + q * rpp;
            float4 a = av[q];
            if (AMODE == ACT_BN_RELU) {
                a.x = relu_np(fmaf(a.x, sc4.x, sh4.x)), a.y = relu_np(fmaf(a.y, sc4.y, sh4.y));
                a.z = relu_np(fmaf(a.z, sc4.z, sh4.z)), a.w = relu_np(fmaf(a.w, sc4.w, sh4.w));
            }
            const float ma = r < R ? 1.f : 0.f;
            a.x *= ma, a.y *= ma, a.z *= ma, a.w *= ma;
            *reinterpret_cast<float4 *>(As + r * LD + c4) = a;
        }
#pragma unroll
    for (int q = 0; q < MAXW; ++q)
        if (q < npw) {
            const int r = r0 + q * rpp;
            float4 w = wv[q];
            const float mw = col0 + r < Co ? 1.f : 0.f;
            w.x *= mw, w.y *= mw, w.z *= mw, w.w *= mw;
            *reinterpret_cast<float4 *>(Ws + r * LD + c4) = w;
        }
    __syncthreads();
    f32x16 acc0, acc1;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc0[e] = 0.f, acc1[e] = 0.f;
    const int kph = Ci / 8, kb = wave * (Ci / 4) + h * kph;
    const float *ap0 = As + l31 * LD + kb, *ap1 = As + (32 + l31) * LD + kb, *bp = Ws + l31 * LD + kb;
    for (int t = 0; t < kph; t += 4) {
        const float4 a0 = *reinterpret_cast<const float4 *>(ap0 + t), a1 = *reinterpret_cast<const float4 *>(ap1 + t);
        const float4 b = *reinterpret_cast<const float4 *>(bp + t);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, b.x, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, b.x, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, b.y, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, b.y, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, b.z, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, b.z, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, b.w, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, b.w, acc1, 0, 0, 0);
    }
    wave_sum_to_wave0(acc0, red);
    wave_sum_to_wave0(acc1, red + 3 * 16 * 64);
    if (wave != 0) return;
    float s0 = 0.f;
    const bool vec_out = Co % 32 == 0;
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = 32 * s2 + frag_row(e, lane);
            const float v = (s2 ? acc1[e] : acc0[e]) + bias;
            if (row < R && colok) {
                s0 += v;
                if (!vec_out) g.z[(size_t)row * Co + col] = v;
            }
            Ts[row * 36 + l31] = v;
        }
    if (vec_out) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = 8 * i + (lane >> 3);
            const float4 v = *reinterpret_cast<const float4 *>(Ts + row * 36 + (lane & 7) * 4);
            if (row < R) *reinterpret_cast<float4 *>(g.z + (size_t)row * Co + col0 + (lane & 7) * 4) = v;
        }
    }
    s0 += __shfl_xor(s0, 32);
    if (g.bn.coef) {  // every row of these 32 columns is here: two-pass variance around the mean (see small_fwd_lds_kernel)
        if (blockIdx.x == 0 && lane == 0 && g.bn.num_batches_tracked) *g.bn.num_batches_tracked += 1;
        const float meanf = s0 / (float)g.bn.R;
        float s2v = 0.f;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float d = ((s2 ? acc1[e] : acc0[e]) + bias) - meanf;
                if (32 * s2 + frag_row(e, lane) < R && colok) s2v += d * d;
            }
        s2v += __shfl_xor(s2v, 32);
        if (lane < 32 && colok) {
            const double rR = fast_rcp((double)g.bn.R);
            const double mean = (double)s0 * rR;
            const double dm = mean - (double)meanf;
            double var = (double)s2v * rR - dm * dm;
            if (var < 0.0) var = 0.0;
            const float invstd = (float)fast_rsqrt(var + (double)g.bn.eps);
            const float sc = bn_g * invstd;
            g.bn.coef[col] = sc;
            g.bn.coef[Co + col] = bn_b - (float)mean * sc;
            g.bn.coef[2 * Co + col] = (float)mean;
            g.bn.coef[3 * Co + col] = invstd;
            if (g.bn.running_mean) {
                const double unbiased = g.bn.R > 1 ? var * (double)g.bn.R * fast_rcp((double)(g.bn.R - 1)) : var;
                g.bn.running_mean[col] = (1.f - g.bn.momentum) * bn_rm + g.bn.momentum * (float)mean;
                g.bn.running_var[col] = (1.f - g.bn.momentum) * bn_rv + g.bn.momentum * (float)unbiased;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// First layer (Ci = 3: the xyz input).  K = 3 is no GEMM: z = w0 x + w1 y + w2 z + b is a streaming kernel bound by
// the 256 B/row it writes; the matrix-core path would spend 95 % of its tile on zero padding.
// Workgroup = 64 rows x 64 output channels (thread = channel, 4 row groups), stats partial per workgroup row block
// exactly like linear_fwd_kernel ([gridDim.x][2][Co], 64 rows per block).
// ------------------------------------------------------------------------------------------------
// Rider of the xyz-layer kernel (statistics-chain stack): the weights of the GEMM layers above it, split once per step into the
// three bf16 planes the split-bf16 GEMMs consume ([3][Co][Ci] per layer, k-contiguous) -- otherwise every one of a layer's 512
// workgroups splits the same W tile again (2/3 of their staging VALU work, which is as long as their MFMA phase).
struct WSplitJob {
    const float *w[4];
    __bf16 *dst[4];
    int elems[4];  // Co * Ci
    int first[5];  // first 1024-element block of layer l in the job's block numbering; first[n] = number of blocks
    int n;
    unsigned long long *zero_keys;  // the last layer's pool keys (FwdArgs::pool_keys), cleared here for this call
    int nkeys;
};
__device__ __forceinline__ void wsplit_block(const WSplitJob &job, int blk, int tid)
{
    int l = 0;
#pragma unroll
    for (int q = 1; q < 4; ++q)
        if (q < job.n && blk >= job.first[q]) l = q;
    const int e = (blk - job.first[l]) * 1024 + tid * 4;
    if (e >= job.elems[l]) return;
    const float4 v = *reinterpret_cast<const float4 *>(job.w[l] + e);
    bf16x4 p1, p2, p3;
    const float f[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        __bf16 h1, h2, h3;
        split3(f[t], h1, h2, h3);
        p1[t] = h1, p2[t] = h2, p3[t] = h3;
    }
    __bf16 *d = job.dst[l] + e;
    *reinterpret_cast<bf16x4 *>(d) = p1;
    *reinterpret_cast<bf16x4 *>(d + job.elems[l]) = p2;
    *reinterpret_cast<bf16x4 *>(d + 2 * (size_t)job.elems[l]) = p3;
}

__global__ void __launch_bounds__(256) conv_in3_fwd_kernel(int R, int Co, const float *__restrict__ x,
                                                           const float *__restrict__ W, const float *__restrict__ bias,
                                                           float *__restrict__ z, float *__restrict__ stats,
                                                           long long *__restrict__ acc_out = nullptr,
                                                           long long *__restrict__ clear_flags = nullptr, WSplitJob job = WSplitJob{},
                                                           int nb = 1)
{
    if (job.n > 0 && blockIdx.y == 0 && (int)blockIdx.x < job.first[job.n]) wsplit_block(job, blockIdx.x, threadIdx.x);
    if (job.zero_keys && blockIdx.y == 0) {
        const int i = blockIdx.x * 256 + threadIdx.x;
        if (i < job.nkeys) job.zero_keys[i] = 0ull;
    }
    // (statistics chain: the poison word of the LAST layer's accumulators, read by every workgroup of the previous
    // step's closing kernel, is reset here, by the first kernel of the next step)
    if (clear_flags && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) clear_flags[0] = 0;
    // thread -> 4 consecutive channels (c4) x row slot rs (16 slots): 16-byte stores, a wave writes 4 whole 256-byte rows
    // per instruction (dword stores cost ~58 issue cycles per wave-instruction: the 16-per-thread version was issue-bound)
    // nb > 1 (statistics chain at large batches): the workgroup walks nb consecutive 64-row blocks and keeps their fixed-point
    // sums in registers -- ONE pair of atomics per channel and workgroup instead of one per 64 rows (at 512 x 1024 points 1 M
    // 64-bit atomics on 2 K addresses paced this kernel at 25 us).  Every block's partial is converted on its own, as fx_add would:
    // the integer totals, hence the statistics, are the same bit for bit.
    __shared__ float red[2][16][64];
    const int q = threadIdx.x & 15, rs = threadIdx.x >> 4;
    const int cl = q * 4, co = blockIdx.y * 64 + cl;
    const bool vec = (Co & 3) == 0 && co + 3 < Co;
    float w[4][3], b[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = min(co + j, Co - 1);
        const float m = co + j < Co ? 1.f : 0.f;
        w[j][0] = W[c * 3 + 0] * m, w[j][1] = W[c * 3 + 1] * m, w[j][2] = W[c * 3 + 2] * m;
        b[j] = bias ? bias[c] * m : 0.f;
    }
    const int nblocks = (R + 63) / 64;
    const int blk0 = blockIdx.x * nb, blk1 = min(blk0 + nb, nblocks);
    long long fx0 = 0, fx1 = 0;
    for (int blk = blk0; blk < blk1; ++blk) {
        const int row0 = blk * 64;
        float xs[4][3];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = min(row0 + rs + 16 * i, R - 1);
            xs[i][0] = x[(size_t)r * 3], xs[i][1] = x[(size_t)r * 3 + 1], xs[i][2] = x[(size_t)r * 3 + 2];
        }
        float s0[4] = {0.f, 0.f, 0.f, 0.f}, s1[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = row0 + rs + 16 * i;
            const float m = r < R ? 1.f : 0.f;
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v[j] = (fmaf(w[j][2], xs[i][2], fmaf(w[j][1], xs[i][1], w[j][0] * xs[i][0])) + b[j]) * m;
                s0[j] += v[j];
                s1[j] += v[j] * v[j];
            }
            if (r < R && z) {  // (z == NULL: statistics only -- the consumers rebuild the activation from the cloud, FwdArgs::x3)
                if (vec) {
                    *reinterpret_cast<float4 *>(z + (size_t)r * Co + co) = make_float4(v[0], v[1], v[2], v[3]);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (co + j < Co) z[(size_t)r * Co + co + j] = v[j];
                }
            }
        }
        if (blk > blk0) __syncthreads();  // (the previous block's sums have been read)
#pragma unroll
        for (int j = 0; j < 4; ++j) red[0][rs][cl + j] = s0[j], red[1][rs][cl + j] = s1[j];
        __syncthreads();
        if (threadIdx.x < 64 && (stats || acc_out)) {
            const int c = blockIdx.y * 64 + threadIdx.x;
            if (c < Co) {
                float a0 = 0.f, a1 = 0.f;
#pragma unroll
                for (int k = 0; k < 16; ++k) a0 += red[0][k][threadIdx.x], a1 += red[1][k][threadIdx.x];
                if (acc_out) {  // fixed-point statistics chain (sn_conv_stack_forward_bn)
                    if (nb == 1) {
                        fx_add<kFxShiftFwd>(acc_out, blockIdx.x % kFxSlots, 0, c, a0);
                        fx_add<kFxShiftFwd>(acc_out, blockIdx.x % kFxSlots, 1, c, a1);
                    } else {
                        fx_local_add<kFxShiftFwd>(fx0, acc_out, 0, c, a0);
                        fx_local_add<kFxShiftFwd>(fx1, acc_out, 1, c, a1);
                    }
                } else {
                    float *st = stats + (size_t)blk * 2 * Co;
                    st[c] = a0;
                    st[Co + c] = a1;
                }
            }
        }
    }
    if (nb > 1 && acc_out && threadIdx.x < 64 && blk0 < blk1) {
        const int c = blockIdx.y * 64 + threadIdx.x, slot = blockIdx.x % kFxSlots;
        if (c < Co) {
            atomicAdd(reinterpret_cast<unsigned long long *>(acc_out + (slot * 2 + 0) * kFxRow + c), (unsigned long long)fx0);
            atomicAdd(reinterpret_cast<unsigned long long *>(acc_out + (slot * 2 + 1) * kFxRow + c), (unsigned long long)fx1);
        }
    }
}

// training: batch statistics from the forward partials -> coef [4][C] = scale, shift, mean, invstd;
// running statistics updated as torch.nn.BatchNorm1d does (unbiased variance, momentum).
__global__ void __launch_bounds__(1024) bn_finalize_kernel(int nblk, int C, const float *__restrict__ stats, BnFwd bn)
{
    if (blockIdx.x == 0 && threadIdx.x == 0 && bn.num_batches_tracked) *bn.num_batches_tracked += 1;
    double s, ss;
    const int c = blockIdx.x * kChan + (threadIdx.x & (kChan - 1));
    BnFwdIn in{};
    if (threadIdx.x < kChan && c < C) in = bn_fwd_inputs(bn, c);  // in flight during the reduction
    if (!partial_sums(nblk, C, stats, blockIdx.x, s, ss)) return;
    bn_finalize_channel(bn, C, c, s, ss, in);
}

// bn_finalize_kernel of the LAST conv layer + the max-pool over the points: the forward epilogue left, per 64-row block and
// channel, the maximum and the minimum of the pre-BN output (pool_val / pool_idx); with the sign of the BatchNorm scale
// known here, pooled[b][c] = relu(scale * (scale >= 0 ? max_n z : min_n z) + shift) is a pick over the cloud's blocks.
// (Replaces a separate 16 MB pass over the layer's output.)
__global__ void __launch_bounds__(1024) bn_finalize_pool_kernel(int nblk, int C, const float *__restrict__ stats, BnFwd bn, int B,
                                                                int bpc, const float *__restrict__ pool_val,
                                                                const int *__restrict__ pool_idx, float *__restrict__ pooled,
                                                                int *__restrict__ argsel, float *__restrict__ zsel,
                                                                long long *__restrict__ acc = nullptr,
                                                                long long *__restrict__ zero_ptr = nullptr, int zero_n = 0)
{
    __shared__ float s_sc[kChan], s_sh[kChan];
    if (blockIdx.x == 0 && threadIdx.x == 0 && bn.num_batches_tracked) *bn.num_batches_tracked += 1;
    double s, ss;
    const int cl = threadIdx.x & (kChan - 1);
    const int c = blockIdx.x * kChan + cl;
    BnFwdIn in{};
    if (threadIdx.x < kChan && c < C) in = bn_fwd_inputs(bn, c);  // in flight during the reduction
    if (acc) {
        // fixed-point statistics chain: the sums are complete in acc [2][C]; this workgroup is the only reader of its
        // channels, so it clears them for the next step; workgroup 0 clears what the previous kernel consumed
        if (threadIdx.x < kChan && c < C) {
            fx_get2<kFxShiftFwd>(acc, c, s, ss);
#pragma unroll
            for (int q = 0; q < kFxSlots * 2 + 2; ++q) acc[q * kFxRow + c] = 0;  // lo rows and the two hi rows
            // the poison word is cleared by the FIRST kernel of the next step (conv_in3_fwd_kernel): every workgroup of this
            // launch must be able to see it
            const float2 cf = bn_finalize_channel(bn, C, c, s, ss, in);
            s_sc[cl] = cf.x, s_sh[cl] = cf.y;
        }
        fx_clear_share(zero_ptr, zero_n, blockIdx.x, gridDim.x, threadIdx.x, 1024);
    } else if (partial_sums(nblk, C, stats, blockIdx.x, s, ss)) {
        const float2 cf = bn_finalize_channel(bn, C, c, s, ss, in);
        s_sc[cl] = cf.x, s_sh[cl] = cf.y;
    }
    __syncthreads();
    if (c >= C) return;
    const float sc = s_sc[cl], sh = s_sh[cl];
    const int sel = sc >= 0.f ? 0 : 1;  // max or min
    // thread -> (channel cl, cloud slot, quarter of the cloud's blocks): the quarter's partials are loaded together, the four
    // quarters (lane bits 3 and 4) are combined by shuffles; blocks hold ascending rows, so on ties the lower index wins
    const int quarter = (threadIdx.x >> 3) & 3, bslot = threadIdx.x >> 5;
    const int per = (bpc + 3) / 4, q0 = quarter * per, q1 = min(bpc, q0 + per);
    for (int b0 = 0; b0 < B; b0 += 32) {
        const int b = b0 + bslot;
        float best = sel == 0 ? -INFINITY : INFINITY;
        int arg = 0x7fffffff;
        if (b < B) {
#pragma unroll 4
            for (int q = q0; q < q1; ++q) {
                const size_t o = ((size_t)(b * bpc + q) * 2 + sel) * C + c;
                const float v = pool_val[o];
                const int i = pool_idx[o];
                if (sel == 0 ? (v > best || (v == best && i < arg)) : (v < best || (v == best && i < arg))) best = v, arg = i;
            }
        }
#pragma unroll
        for (int ofs = 8; ofs <= 16; ofs <<= 1) {
            const float ob = __shfl_xor(best, ofs);
            const int oa = __shfl_xor(arg, ofs);
            if (sel == 0 ? (ob > best || (ob == best && oa < arg)) : (ob < best || (ob == best && oa < arg))) best = ob, arg = oa;
        }
        if (quarter == 0 && b < B) {
            pooled[(size_t)b * C + c] = relu_np(fmaf(best, sc, sh));
            argsel[(size_t)b * C + c] = arg;
            zsel[(size_t)b * C + c] = best;
        }
    }
}

// bn_finalize_pool_kernel for any batch: the last conv layer left, per cloud and channel, (max Z, first row) / (min Z, first row) as
// 64-bit keys (FwdArgs::pool_keys) instead of per-64-row-block partials, so the pick is a decode, spread over the clouds (the
// block-partial kernel reads B * N / 64 partials per channel with four workgroups: 6 us at B = 32, 50 us at B = 512).  Every
// workgroup finalises the BatchNorm of all C <= 128 channels itself from the fixed-point sums (35 words per channel); workgroup 0
// stores the coefficients / running statistics; the workgroup that arrives last at the counter word behind the poison word
// clears the sums (every workgroup has read them by then); the accumulators the previous kernel consumed are cleared in shares.
// Same expressions as the FC chain's pool stage: bit-identical pooled / argsel / zsel.
constexpr int kFxArrive = kFxPoison + 8;  // spare word of a layer's accumulator block
__global__ void __launch_bounds__(256) bn_finalize_pool_keys_kernel(BnFwd bn, int B, int C, int cpb,
                                                                    const unsigned long long *__restrict__ keys,
                                                                    float *__restrict__ pooled, int *__restrict__ argsel,
                                                                    float *__restrict__ zsel, long long *__restrict__ acc,
                                                                    long long *__restrict__ zero_ptr, int zero_n)
{
    __shared__ float s_sc[128], s_sh[128];
    __shared__ int s_last;
    const int tid = threadIdx.x;
    if (blockIdx.x == 0 && tid == 0 && bn.num_batches_tracked) *bn.num_batches_tracked += 1;
    if (tid < C) {
        const BnFwdIn in = bn_fwd_inputs(bn, tid);
        double s, ss;
        fx_get2<kFxShiftFwd>(acc, tid, s, ss);
        const float2 cf = bn_finalize_channel(bn, C, tid, s, ss, in, blockIdx.x == 0);
        s_sc[tid] = cf.x, s_sh[tid] = cf.y;
    }
    fx_clear_share(zero_ptr, zero_n, blockIdx.x, gridDim.x, tid, 256);
    __syncthreads();  // (every read of acc by this workgroup has returned)
    if (tid == 0)
        s_last = atomicAdd(reinterpret_cast<unsigned long long *>(acc + kFxArrive), 1ull) == (unsigned long long)gridDim.x - 1;
    const int b0 = blockIdx.x * cpb, n = min(cpb, B - b0) * C;
    for (int i = tid; i < n; i += 256) {
        const int b = b0 + i / C, c = i % C;
        const float sc = s_sc[c], sh = s_sh[c];
        float v;
        int row;
        if (sc >= 0.f) {
            pool_key_decode(keys[((size_t)b * 2) * C + c], v, row);
        } else {
            pool_key_decode(keys[((size_t)b * 2 + 1) * C + c], v, row);
            v = -v;
        }
        const size_t o = (size_t)b * C + c;
        pooled[o] = relu_np(fmaf(v, sc, sh)), argsel[o] = row, zsel[o] = v;
    }
    __syncthreads();
    if (s_last)  // lo rows, hi rows and the arrival word; the poison word belongs to the first kernel of the next step
        for (int i = tid; i < kFxLayer; i += 256)
            if (i != kFxPoison) acc[i] = 0;
}

// Batch statistics of a SHORT activation matrix (the FC head at batches above 32: R rows, a few hundred at most) straight from
// z in two passes -- mean, then the squares around it, in double.  Behind the max-pool the head's features are nearly the same
// for every cloud (|mean| / std of 10..100): E[z^2] - mean^2 from fp32 block sums loses 2..4 digits of the variance there.
__global__ void __launch_bounds__(1024) bn_twopass_kernel(int R, int C, const float *__restrict__ z, BnFwd bn)
{
    // 64 channels x 16 row stripes per workgroup, eight independent loads in flight per thread (round 4: four stripes and one
    // dependent load per trip took 46 us for 512 x 256 values -- a memory round trip per row); fixed summation order
    __shared__ double red[16][64];
    if (blockIdx.x == 0 && threadIdx.x == 0 && bn.num_batches_tracked) *bn.num_batches_tracked += 1;
    const int lane = threadIdx.x & 63, stripe = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const bool ok = c < C;
    BnFwdIn in{};
    if (ok && stripe == 0) in = bn_fwd_inputs(bn, c);
    const float *zc = z + (ok ? c : 0);
    const auto total = [&]() {
        double t = 0.0;
#pragma unroll
        for (int q = 0; q < 16; ++q) t += red[q][lane];
        return t;
    };
    double s = 0.0;
    int r = stripe;
    for (; r + 7 * 16 < R; r += 8 * 16) {
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = zc[(size_t)(r + 16 * i) * C];
#pragma unroll
        for (int i = 0; i < 8; ++i) s += (double)v[i];
    }
    for (; r < R; r += 16) s += (double)zc[(size_t)r * C];
    red[stripe][lane] = s;
    __syncthreads();
    const double mean = total() / (double)R;
    __syncthreads();
    double q = 0.0;
    r = stripe;
    for (; r + 7 * 16 < R; r += 8 * 16) {
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = zc[(size_t)(r + 16 * i) * C];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const double d = (double)v[i] - mean;
            q += d * d;
        }
    }
    for (; r < R; r += 16) {
        const double d = (double)zc[(size_t)r * C] - mean;
        q += d * d;
    }
    red[stripe][lane] = q;
    __syncthreads();
    if (stripe == 0 && ok) bn_finalize_channel_mv(bn, C, c, mean, total() / (double)R, in);
}

// y = z scale + shift, no activation: the BatchNorm on the head's OUTPUT (classification sampler) where the GEMM kernel could
// not apply it in its epilogue (more than 32 rows: statistics from bn_twopass_kernel; eval mode: running statistics)
__global__ void __launch_bounds__(256) bn_apply_kernel(long long n4, int C, const float *__restrict__ z, const float *__restrict__ coef,
                                                       float *__restrict__ y)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;  // float4 index; C % 4 == 0
    if (i >= n4) return;
    const int c = (int)((i * 4) % C);
    const float4 v = reinterpret_cast<const float4 *>(z)[i];
    const float4 sc = *reinterpret_cast<const float4 *>(coef + c), sh = *reinterpret_cast<const float4 *>(coef + C + c);
    reinterpret_cast<float4 *>(y)[i] = make_float4(fmaf(v.x, sc.x, sh.x), fmaf(v.y, sc.y, sh.y), fmaf(v.z, sc.z, sh.z), fmaf(v.w, sc.w, sh.w));
}

// Backward of that output BatchNorm: gy (R, C) upstream gradient, z (R, C) its input, coef (4, C) of the forward ->
// dz (R, C), dgamma, dbeta (C).  Batch statistics (fixed == 0):  dbeta = sum gy, dgamma = invstd sum gy (z - mean),
// dz = k1 gy + k2 z + k3 with k1 = scale, k2 = -scale invstd dgamma / R, k3 = scale (invstd mean dgamma / R - dbeta / R) -- the
// expressions of the GEMM kernels' BatchNorm-backward epilogues (fc_chain_bwd_kernel), without a ReLU mask.  Running statistics
// (fixed != 0, an eval-mode forward): dz = scale gy.  64 channels x 16 row stripes per workgroup, sums in double, fixed order.
__global__ void __launch_bounds__(1024) bn_output_bwd_kernel(int R, int C, int fixed, const float *__restrict__ gy,
                                                             const float *__restrict__ z, const float *__restrict__ coef,
                                                             float *__restrict__ dz, float *__restrict__ dgamma, float *__restrict__ dbeta)
{
    __shared__ double red[2][16][64];
    __shared__ float ks[3][64];
    const int lane = threadIdx.x & 63, stripe = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const bool ok = c < C;
    const int cc = ok ? c : 0;
    const float scale = coef[cc], mean = coef[2 * C + cc], invstd = coef[3 * C + cc];
    double s0 = 0.0, s1 = 0.0;
    for (int r = stripe; r < R; r += 16) {
        const float g = gy[(size_t)r * C + cc], v = z[(size_t)r * C + cc];
        s0 += (double)g;
        s1 += (double)g * (double)(v - mean);
    }
    red[0][stripe][lane] = s0, red[1][stripe][lane] = s1;
    __syncthreads();
    if (stripe == 0) {
        double t0 = 0.0, t1 = 0.0;
#pragma unroll
        for (int q = 0; q < 16; ++q) t0 += red[0][q][lane], t1 += red[1][q][lane];
        const double dg = (double)invstd * t1, rinv = 1.0 / (double)R, sc = scale;
        if (ok) dgamma[c] = (float)dg, dbeta[c] = (float)t0;
        ks[0][lane] = scale;
        ks[1][lane] = fixed ? 0.f : (float)(-sc * (double)invstd * dg * rinv);
        ks[2][lane] = fixed ? 0.f : (float)(sc * ((double)invstd * (double)mean * dg * rinv - t0 * rinv));
    }
    __syncthreads();
    const float k1 = ks[0][lane], k2 = ks[1][lane], k3 = ks[2][lane];
    if (ok)
        for (int r = stripe; r < R; r += 16) {
            const size_t o = (size_t)r * C + c;
            dz[o] = fmaf(k1, gy[o], fmaf(k2, z[o], k3));
        }
}

// eval: coefficients from the running statistics
__global__ void bn_eval_coef_kernel(int C, const float *__restrict__ gamma, const float *__restrict__ beta, float eps,
                                    const float *__restrict__ running_mean, const float *__restrict__ running_var,
                                    float *__restrict__ coef)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float invstd = 1.0f / sqrtf(running_var[c] + eps);
    const float sc = gamma[c] * invstd;
    coef[c] = sc;
    coef[C + c] = beta[c] - running_mean[c] * sc;
    coef[2 * C + c] = running_mean[c];
    coef[3 * C + c] = invstd;
}

// ------------------------------------------------------------------------------------------------
// max pooling over the points of each cloud, fused with BN + ReLU of the last conv layer
//   pooled[b][c] = max_n relu(scale z + shift) = relu(scale * (scale >= 0 ? max_n z : min_n z) + shift)
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024) pool_fwd_kernel(int N, int C, const float *__restrict__ z,
                                                        const float *__restrict__ coef, float *__restrict__ pooled,
                                                        int *__restrict__ argsel, float *__restrict__ zsel)
{
    constexpr int NW = 16;
    __shared__ float smax[NW][64], smin[NW][64];
    __shared__ int amax[NW][64], amin[NW][64];
    const int b = blockIdx.x, cl = threadIdx.x & 63, c = blockIdx.y * 64 + cl;
    const int w = threadIdx.x >> 6;
    float vmax = -INFINITY, vmin = INFINITY;
    int imax = 0, imin = 0;
    if (c < C) {
        const float *zb = z + (size_t)b * N * C + c;
        // wave w scans rows w, w+16, ... in ascending order: the first occurrence wins within the wave
        int n = w;
        for (; n + 3 * NW < N; n += 4 * NW) {
            const float v0 = zb[(size_t)n * C], v1 = zb[(size_t)(n + NW) * C], v2 = zb[(size_t)(n + 2 * NW) * C],
                        v3 = zb[(size_t)(n + 3 * NW) * C];
            if (v0 > vmax) vmax = v0, imax = n;
            if (v0 < vmin) vmin = v0, imin = n;
            if (v1 > vmax) vmax = v1, imax = n + NW;
            if (v1 < vmin) vmin = v1, imin = n + NW;
            if (v2 > vmax) vmax = v2, imax = n + 2 * NW;
            if (v2 < vmin) vmin = v2, imin = n + 2 * NW;
            if (v3 > vmax) vmax = v3, imax = n + 3 * NW;
            if (v3 < vmin) vmin = v3, imin = n + 3 * NW;
        }
        for (; n < N; n += NW) {
            const float v = zb[(size_t)n * C];
            if (v > vmax) vmax = v, imax = n;
            if (v < vmin) vmin = v, imin = n;
        }
    }
    smax[w][cl] = vmax, smin[w][cl] = vmin;
    amax[w][cl] = imax, amin[w][cl] = imin;
    __syncthreads();
    if (w == 0 && c < C) {
        for (int q = 1; q < NW; ++q) {  // ties across waves: lowest row index
            const float a = smax[q][cl], bb = smin[q][cl];
            const int ia = amax[q][cl], ib = amin[q][cl];
            if (a > vmax || (a == vmax && ia < imax)) vmax = a, imax = ia;
            if (bb < vmin || (bb == vmin && ib < imin)) vmin = bb, imin = ib;
        }
        const float sc = coef[c], sh = coef[C + c];
        const bool up = sc >= 0.f;
        const float zs = up ? vmax : vmin;
        pooled[(size_t)b * C + c] = relu_np(fmaf(zs, sc, sh));
        argsel[(size_t)b * C + c] = up ? imax : imin;
        zsel[(size_t)b * C + c] = zs;
    }
}

}  // namespace sn

template <int AMODE>
static void launch_fwd(const FwdArgs &g, hipStream_t st, bool few_rows = false)
{
    const int R = g.a.rows, Ci = g.w.ci, Co = g.w.co;
    if (AMODE == ACT_NONE && Ci == 3 && R > 64) {  // xyz input layer: streaming kernel, same stats layout (64 rows / block)
        static_assert(TileBig::BM == 64, "conv_in3_fwd_kernel writes one stats partial per 64 rows");
        hipLaunchKernelGGL(conv_in3_fwd_kernel, dim3((R + 63) / 64, (Co + 63) / 64), dim3(256), 0, st, R, Co, g.a.z, g.w.w,
                           g.bias, g.z, g.stats);
        return;
    }
    // few_rows (sn_linear_forward_rows), 32 < R, a layer of the FC head at a mid-size batch: the tile kernels below put (R / 64) x (Co / 64) workgroups on the chip
    // (32 at R = 512, Co = 256), each walking K in dependent 64-wide chunks -- 15 us of exposed load latency for 0.07 GFLOP.  While
    // (R / 32) x (Co / 32) workgroups fit the chip in one wave, the R <= 32 kernel runs them row block by row block instead: both
    // operands of a workgroup (32 rows, 32 columns, all of K) are in flight at once.  No statistics: the caller takes them from Z
    // (sn_bn_batch_stats_twopass), as the FC head does above 32 rows.
    // (small_fwd_lds_kernel / rows64_fwd_kernel stage a row as Ci / 4 threads x 16 bytes: Ci / 4 must divide the 256 threads --
    //  64, 128, 256, 512; at 192 / 320 / 384 / 448 the staging left rows half-filled: wrong outputs until round 4)
    const bool lds_ci = Ci >= 64 && Ci <= 512 && (Ci & (Ci - 1)) == 0;
    const bool row_tiled = few_rows && R > 32 && lds_ci && !g.stats && !g.bn.coef && !g.pool_val && !g.pool_keys && !g.wplanes &&
                           AMODE != ACT_BN_RELU_FX && (long long)((R + 31) / 32) * ((Co + 31) / 32) <= device_cus();
    if (R <= 32 || row_tiled) {
        const dim3 grid((Co + 31) / 32, (R + 31) / 32);
        if (lds_ci) {
            const size_t lds = ((size_t)64 * (Ci + 4) + 3 * 16 * 64 + 32 * 36) * sizeof(float);
            static SnLdsAttr attr;
            // (a refused request leaves the error text; the launch then fails and the entry point's launch check reports it)
            (void)sn_lds_attr(attr, (const void *)small_fwd_lds_kernel<AMODE>, ((size_t)64 * 516 + 3 * 16 * 64 + 32 * 36) * sizeof(float),
                              "small_fwd_lds_kernel");
            hipLaunchKernelGGL((small_fwd_lds_kernel<AMODE>), grid, dim3(256), lds, st, g);
        } else if (Ci % 64 == 0)
            hipLaunchKernelGGL((small_fwd_kernel<AMODE, true>), dim3((Co + 31) / 32), dim3(256), 0, st, g);
        else
            hipLaunchKernelGGL((small_fwd_kernel<AMODE, false>), dim3((Co + 31) / 32), dim3(256), 0, st, g);
    } else if (R > 64) {
        dim3 grid((R + TileBig::BM - 1) / TileBig::BM, (Co + TileBig::BN - 1) / TileBig::BN);
        const bool full = R % TileBig::BM == 0 && Co % TileBig::BN == 0 && Ci % BK == 0;
        SN_LAUNCH_T(linear_fwd_kernel, TileBig, full, grid, g, AMODE);
    } else {
        dim3 grid((R + TileSmall::BM - 1) / TileSmall::BM, (Co + TileSmall::BN - 1) / TileSmall::BN);
        const bool full = R % TileSmall::BM == 0 && Co % TileSmall::BN == 0 && Ci % BK == 0;
        SN_LAUNCH_T(linear_fwd_kernel, TileSmall, full, grid, g, AMODE);
    }
}

extern "C" int sn_linear_forward(int R, int Ci, int Co, const float *ain, const float *coef_prev, const float *W,
                                 const float *bias, float *z, float *stats, sn_stream_t stream)
{
    SN_REQUIRE(R >= 1 && Ci >= 1 && Co >= 1, "bad size");
    SN_REQUIRE(ain && W && z, "null pointer");
    FwdArgs g{};
    g.a = make_act(ain, coef_prev, R, Ci);
    g.w.w = W, g.w.co = Co, g.w.ci = Ci;
    g.bias = bias, g.z = z, g.stats = stats;
    hipStream_t st = (hipStream_t)stream;
    if (coef_prev)
        launch_fwd<ACT_BN_RELU>(g, st);
    else
        launch_fwd<ACT_NONE>(g, st);
    SN_LAUNCH_CHECK();
    return 0;
}

extern "C" int sn_linear_forward_rows(int R, int Ci, int Co, const float *ain, const float *coef_prev, const float *W, const float *bias,
                                      float *z, sn_stream_t stream)
{
    SN_REQUIRE(R >= 1 && Ci >= 1 && Co >= 1, "bad size");
    SN_REQUIRE(ain && W && z, "null pointer");
    FwdArgs g{};
    g.a = make_act(ain, coef_prev, R, Ci);
    g.w.w = W, g.w.co = Co, g.w.ci = Ci;
    g.bias = bias, g.z = z, g.stats = nullptr;
    if (coef_prev)
        launch_fwd<ACT_BN_RELU>(g, (hipStream_t)stream, true);
    else
        launch_fwd<ACT_NONE>(g, (hipStream_t)stream, true);
    SN_LAUNCH_CHECK();
    return 0;
}

// ---- last layer of a BatchNorm-free ReLU stack + max over the points, WITHOUT its activation tensor (PCRNet's PointNetFeatures,
// registration/models/pcrnet.py:23-41: the 128 -> 1024 layer on 32 x 1024 points is 134 MB that the reference writes, reads back
// for the max and -- for the frozen template branch -- never needs again).  The GEMM's epilogue combines, per cloud and channel,
// (max Z, first row) as 64-bit keys by atomicMax (order-independent); a small kernel decodes pooled = relu(max), the row and
// the pre-activation value (what the pooling backward needs).  keys: B * 2 * Co u64 of scratch (cleared here).
__global__ void __launch_bounds__(256) maxpool_keys_decode_kernel(int n, int Co, const unsigned long long *__restrict__ keys,
                                                                  float *__restrict__ pooled, int *__restrict__ argsel,
                                                                  float *__restrict__ zsel)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v;
    int row;
    pool_key_decode(keys[(size_t)(i / Co) * 2 * Co + i % Co], v, row);  // keys [B][2][Co]: the maxima are plane 0 (FwdArgs::pool_keys)
    pooled[i] = relu_np(v);
    if (argsel) argsel[i] = row;
    if (zsel) zsel[i] = v;
}

extern "C" int sn_linear_forward_maxpool_supported(int R, int Ci, int Co, int npts)
{
    return R > 64 && npts >= 64 && npts % 64 == 0 && R % npts == 0 && R % TileBig::BM == 0 && Co % TileBig::BN == 0 && Ci % BK == 0;
}

extern "C" int sn_linear_forward_maxpool(int R, int Ci, int Co, int npts, const float *ain, const float *coef_prev, const float *W,
                                         const float *bias, float *z, unsigned long long *keys, float *pooled, int *argsel,
                                         float *zsel, int keys_cleared, sn_stream_t stream)
{
    SN_REQUIRE(R >= 1 && Ci >= 1 && Co >= 1 && npts >= 1, "bad size");
    SN_REQUIRE(ain && W && keys && pooled && coef_prev, "null pointer");
    if (!sn_linear_forward_maxpool_supported(R, Ci, Co, npts))
        return sn_set_error(SN_ERR_UNSUPPORTED, "sn_linear_forward_maxpool: needs 64-aligned rows per cloud / channels");
    hipStream_t st = (hipStream_t)stream;
    const int B = R / npts;
    if (!keys_cleared) {  // (keys_cleared: an earlier launch on the stream zeroed them -- sn_pointnet_narrow_forward's rider)
        const hipError_t e = hipMemsetAsync(keys, 0, (size_t)B * 2 * Co * sizeof(unsigned long long), st);
        if (e != hipSuccess) return sn_set_error((int)e, "%s: %s", __func__, hipGetErrorString(e));
    }
    FwdArgs g{};
    g.a = make_act(ain, coef_prev, R, Ci);
    g.w.w = W, g.w.co = Co, g.w.ci = Ci;
    g.bias = bias, g.z = z, g.stats = nullptr;  // z == NULL: the activations are not written at all
    g.pool_keys = keys, g.pool_npts = npts, g.pool_max_only = 1;
    launch_fwd<ACT_BN_RELU>(g, st);
    hipLaunchKernelGGL(maxpool_keys_decode_kernel, dim3((B * Co + 255) / 256), dim3(256), 0, st, B * Co, Co, keys, pooled, argsel, zsel);
    SN_LAUNCH_CHECK();
    return 0;
}

extern "C" int sn_linear_stats_blocks(int R);

// Layer forward INCLUDING its BatchNorm finalisation (training): Z, then coef[4][Co] + running statistics.
// R <= 32: one launch (the epilogue finalises); otherwise the GEMM launch + bn_finalize.  stats: scratch of
// sn_linear_stats_blocks(R) * 2 * Co floats.
extern "C" int sn_layer_forward_bn(int R, int Ci, int Co, const float *ain, const float *coef_prev, const float *W,
                                   const float *bias, float *z, float *stats, const float *gamma, const float *beta,
                                   float eps, float momentum, float *running_mean, float *running_var,
                                   long long *num_batches_tracked, float *coef, sn_stream_t stream)
{
    SN_REQUIRE(R >= 1 && Ci >= 1 && Co >= 1, "bad size");
    SN_REQUIRE(ain && W && z && stats && gamma && beta && coef, "null pointer");
    FwdArgs g{};
    g.a = make_act(ain, coef_prev, R, Ci);
    g.w.w = W, g.w.co = Co, g.w.ci = Ci;
    g.bias = bias, g.z = z, g.stats = stats;
    const BnFwd bn{gamma, beta, running_mean, running_var, num_batches_tracked, coef, eps, momentum, (long long)R};
    hipStream_t st = (hipStream_t)stream;
    if (R > 32 && R <= 64 && (Ci == 64 || Ci == 128 || Ci == 256)) {  // both halves in one workgroup, BatchNorm finalised in the epilogue
        g.bn = bn;
        g.stats = nullptr;
        const size_t lds = ((size_t)96 * (Ci + 4) + 2 * 3 * 16 * 64 + 64 * 36) * sizeof(float);
        static SnLdsAttr attr[2];
        const size_t mx = ((size_t)96 * 260 + 2 * 3 * 16 * 64 + 64 * 36) * sizeof(float);
        if (sn_lds_attr(attr[coef_prev ? 1 : 0], coef_prev ? (const void *)rows64_fwd_kernel<ACT_BN_RELU> : (const void *)rows64_fwd_kernel<ACT_NONE>,
                        mx, "rows64_fwd_kernel"))
            return SN_ERR_UNSUPPORTED;
        if (coef_prev)
            hipLaunchKernelGGL((rows64_fwd_kernel<ACT_BN_RELU>), dim3((Co + 31) / 32), dim3(256), lds, st, g);
        else
            hipLaunchKernelGGL((rows64_fwd_kernel<ACT_NONE>), dim3((Co + 31) / 32), dim3(256), lds, st, g);
        SN_LAUNCH_CHECK();
        return 0;
    }
    if (R <= 32) {
        g.bn = bn;
        g.stats = nullptr;
    }
    if (coef_prev)
        launch_fwd<ACT_BN_RELU>(g, st);
    else
        launch_fwd<ACT_NONE>(g, st);
    if (R > 32)
        hipLaunchKernelGGL(bn_finalize_kernel, dim3((Co + kChan - 1) / kChan), dim3(1024), 0, st, sn_linear_stats_blocks(R), Co, stats, bn);
    SN_LAUNCH_CHECK();
    return 0;
}

// sn_layer_forward_bn for the last conv layer, with the max-pool over the npts points of every cloud folded in:
// pooled / argsel / zsel as sn_pool_forward.  pool_val (floats) / pool_idx (ints): scratch of sn_linear_stats_blocks(R)*2*Co
// elements each.  Needs R % 64 == 0, npts % 64 == 0, Co % 64 == 0, Ci % 64 == 0 (else SN_ERR_UNSUPPORTED: call
// sn_layer_forward_bn + sn_pool_forward).
extern "C" int sn_conv_forward_bn_pool(int R, int Ci, int Co, int npts, const float *ain, const float *coef_prev, const float *W,
                                       const float *bias, float *z, float *stats, const float *gamma, const float *beta,
                                       float eps, float momentum, float *running_mean, float *running_var,
                                       long long *num_batches_tracked, float *coef, float *pool_val, int *pool_idx,
                                       float *pooled, int *argsel, float *zsel, sn_stream_t stream)
{
    SN_REQUIRE(R >= 1 && Ci >= 1 && Co >= 1 && npts >= 1, "bad size");
    SN_REQUIRE(ain && W && z && stats && gamma && beta && coef && pool_val && pool_idx && pooled && argsel && zsel && coef_prev,
               "null pointer");
    if (R <= 64 || R % 64 || npts % 64 || R % npts || Co % 64 || Ci % 64)
        return sn_set_error(SN_ERR_UNSUPPORTED, "sn_conv_forward_bn_pool: needs 64-aligned rows / points / channels");
    FwdArgs g{};
    g.a = make_act(ain, coef_prev, R, Ci);
    g.w.w = W, g.w.co = Co, g.w.ci = Ci;
    g.bias = bias, g.z = z, g.stats = stats;
    g.pool_val = pool_val, g.pool_idx = pool_idx, g.pool_npts = npts;
    const BnFwd bn{gamma, beta, running_mean, running_var, num_batches_tracked, coef, eps, momentum, (long long)R};
    hipStream_t st = (hipStream_t)stream;
    launch_fwd<ACT_BN_RELU>(g, st);
    hipLaunchKernelGGL(bn_finalize_pool_kernel, dim3((Co + kChan - 1) / kChan), dim3(1024), 0, st, sn_linear_stats_blocks(R), Co,
                       stats, bn, R / npts, npts / 64, pool_val, pool_idx, pooled, argsel, zsel);
    SN_LAUNCH_CHECK();
    return 0;
}

// The whole training-mode conv stack of the PointNet head (samplenet.py:90-95: conv/bn/relu x nlayers on (B, 3, N), then
// the max over the points) as ONE call and nlayers + 1 launches: xyz layer, nlayers - 1 GEMM layers, pool / last-BatchNorm
// finalisation.  Batch statistics travel as fixed-point sums (fx_add): each GEMM finalises the BatchNorm of its INPUT in
// its prologue, so no reduction launch sits between two layers.  channels [nlayers + 1] = 3, C1, ..., Cn.
// acc: persistent scratch of sn_conv_stack_acc_elems(nlayers) long long, ZERO before the first call; every call leaves it zero again.
// Per layer l: W[l] (C_{l+1}, C_l), bias[l], gamma / beta / running_mean / running_var [l] (C_{l+1}), z[l] (B*N, C_{l+1})
// pre-BatchNorm output, coef[l] (4, C_{l+1}).  Outputs pooled / zsel (B, Cn), argsel (B, Cn) as sn_pool_forward.
extern "C" int sn_conv_stack_forward_supported(int B, int N, int nlayers, const int *channels)
{
    if (B < 1 || N < 1 || nlayers < 2 || !channels || channels[0] != 3) return 0;
    const long long R = (long long)B * N;
    if (R <= 64 || R > (1ll << 30) || N % 64) return 0;
    bool wide = false;
    long long nb = 0;
    for (int l = 1; l <= nlayers; ++l) {
        if (channels[l] % 64 || channels[l] > 2 * kFxRow) return 0;
        wide = wide || channels[l] > 128;
        if (l < nlayers) nb += ((long long)channels[l] * channels[l + 1] + 1023) / 1024;
    }
    // a layer wider than 128 channels (reconstruction/src/samplers.py:23-38: 64-128-128-256-bottleneck) runs on the pre-split
    // weight planes only (K = 256 known at compile time): the split must fit the xyz layer's workgroups, the pool keys 128 channels
    if (wide && (!SN_BF16X3 || nlayers - 1 > 4 || nb > R / 64 || channels[nlayers] > 128 || channels[1] > 128)) return 0;
    return 1;
}

// (behind the accumulators: room for the split weights of the nlayers - 1 GEMM layers, 128 x 128 x 3 bf16 each -- scratch of the
//  call, contents irrelevant between calls)
[[maybe_unused]] constexpr long long kWPlaneLL = (long long)128 * 128 * 3 * 2 / 8;
// a stack with a layer above 128 channels ("wide"): two accumulator blocks per layer (back to back: one clear range), planes of
// 256 x 256 x 3 bf16 per layer.  The stacks of 64 / 128 channels keep the layout they always had.
[[maybe_unused]] constexpr long long kWPlaneWideLL = (long long)256 * 256 * 3 * 2 / 8;
static bool conv_stack_wide(int nlayers, const int *channels)
{
    for (int l = 1; l <= nlayers; ++l)
        if (channels[l] > kFxRow) return true;
    return false;
}
// 1: sn_conv_stack_forward_bn accepts z[0] == NULL for this shape (the xyz layer's activation is not materialised; conv2's
// forward and sn_conv_stack_backward rebuild it from the cloud): 3 -> 64 -> 64 channels and enough row blocks for the weight split
extern "C" int sn_conv_stack_z1_free_supported(int B, int N, int nlayers, const int *channels)
{
#if SN_BF16X3
    if (!sn_conv_stack_forward_supported(B, N, nlayers, channels) || nlayers < 3 || nlayers - 1 > 4) return 0;
    if (channels[1] != 64 || channels[2] != 64) return 0;
    long long nb = 0;
    for (int l = 1; l < nlayers; ++l) nb += ((long long)channels[l] * channels[l + 1] + 1023) / 1024;
    return nb <= (long long)B * N / 64 ? 1 : 0;
#else
    return 0;
#endif
}
// the statistics region at the head of acc: one block of sums per layer, TWO when a layer of the stack is wider than 128 channels
// (channels == NULL: the narrow layout).  Behind it sits the forward's scratch (split-weight planes), which is NOT zero between calls.
extern "C" long long sn_conv_stack_acc_sum_elems(int nlayers, const int *channels)
{
    if (nlayers <= 0) return 0;
    return (long long)nlayers * ((channels && conv_stack_wide(nlayers, channels)) ? 2 : 1) * kFxLayer;
}
extern "C" long long sn_conv_stack_acc_elems(int nlayers)
{
    // (sized for the wide layout: the caller allocates before it knows the channels)
    return nlayers > 0 ? (long long)nlayers * 2 * kFxLayer + (long long)(nlayers - 1) * kWPlaneWideLL : 0;
}

// tiles per workgroup from which the conv stack's forward GEMMs run as persistent kernels (0: never); a test / A-B hook
static int g_persist_min_tiles = 4;
// 64-row blocks per workgroup of the xyz layer's statistics pass (0: chosen per call, see sn_conv_stack_forward_bn); a test hook
static int g_in3_blocks = 0;
extern "C" int sn_conv_stack_set_in3_blocks(int blocks)
{
    const int prev = g_in3_blocks;
    g_in3_blocks = blocks < 0 ? 0 : blocks;
    return prev;
}
extern "C" int sn_conv_stack_set_persist_min_tiles(int tiles)
{
    const int old = g_persist_min_tiles;
    g_persist_min_tiles = tiles;
    return old;
}
#ifndef SN_FWD_HT
#define SN_FWD_HT 1
#endif
template <class TT, int KT, bool IN3A>
static int launch_fwd_persist(const FwdArgs &g, int ntiles, int tpw, int nwg, hipStream_t st)
{
    const size_t lds = sizeof(float) * ((size_t)(KT / BKX) * 3 * TT::BM * LDX + 2 * KT + (size_t)2 * TT::WR * 2 * TT::BN + (size_t)TT::WR * TT::WC * 16 * 36);
    static SnLdsAttr attr;
    if (sn_lds_attr(attr, (const void *)linear_fwd_persist_kernel<TT, KT, IN3A, 2>, lds, "linear_fwd_persist_kernel")) return SN_ERR_UNSUPPORTED;
    hipLaunchKernelGGL((linear_fwd_persist_kernel<TT, KT, IN3A, 2>), dim3(nwg), dim3(TT::THREADS), lds, st, g, ntiles, tpw);
    return 0;
}

extern "C" int sn_conv_stack_forward_bn(int B, int N, int nlayers, const int *channels, const float *x, const float *const *W,
                                        const float *const *bias, const float *const *gamma, const float *const *beta,
                                        float *const *running_mean, float *const *running_var,
                                        long long *const *num_batches_tracked, const float *eps, const float *momentum,
                                        float *const *z, float *const *coef, long long *acc, float *pool_val, int *pool_idx,
                                        float *pooled, int *argsel, float *zsel, sn_stream_t stream)
{
    if (!sn_conv_stack_forward_supported(B, N, nlayers, channels))
        return sn_set_error(SN_ERR_UNSUPPORTED, "sn_conv_stack_forward_bn: needs N % 64 == 0 and 64 / 128 (/ 256 inside the stack) channels");
    SN_REQUIRE(x && W && gamma && beta && eps && momentum && z && coef && acc && pool_val && pool_idx, "null pointer");
    SN_REQUIRE((pooled && argsel && zsel) || (!pooled && !argsel && !zsel), "pooled / argsel / zsel: all or none");
    hipStream_t st = (hipStream_t)stream;
    const int R = B * N;
    // accumulator blocks per layer: one, or two back to back when a layer has more than 128 channels (then for every layer: one stride)
    const bool wide = conv_stack_wide(nlayers, channels);
    const size_t S = wide ? 2 * (size_t)kFxLayer : (size_t)kFxLayer;
    SN_REQUIRE(!wide || pooled, "a stack with a layer above 128 channels finishes its own pool (the FC chain's pool stage reads the 128-channel layout)");
    auto bn_of = [&](int l) {
        return BnFwd{gamma[l], beta[l], running_mean ? running_mean[l] : nullptr, running_var ? running_var[l] : nullptr,
                     num_batches_tracked ? num_batches_tracked[l] : nullptr, coef[l], eps[l], momentum[l], (long long)R};
    };
    for (int l = 0; l < nlayers; ++l) SN_REQUIRE(W[l] && gamma[l] && beta[l] && (z[l] || l == 0) && coef[l], "null pointer");
    // z[0] == NULL: the first activation tensor is not materialised (its consumers rebuild it from the cloud)
    const bool z1free = z[0] == nullptr;
#if !SN_BF16X3
    SN_REQUIRE(!z1free, "z[0] == NULL needs the split-bf16 build");
#endif
    SN_REQUIRE(!z1free || (channels[1] == 64 && channels[2] == 64), "z[0] == NULL: 3 -> 64 -> 64 channels only");
    // layer 0: xyz input (+ the weight split of the layers above, SN_BF16X3)
    WSplitJob job{};
    __bf16 *planes[8] = {nullptr};
#if SN_BF16X3
    if (nlayers - 1 <= 4) {
        __bf16 *base = reinterpret_cast<__bf16 *>(acc + (size_t)nlayers * S);
        int nb = 0;
        for (int l = 1; l < nlayers; ++l) {
            const int li = l - 1, el = channels[l] * channels[l + 1];
            planes[l] = base + (size_t)li * (wide ? kWPlaneWideLL : kWPlaneLL) * 4;
            job.w[li] = W[l], job.dst[li] = planes[l], job.elems[li] = el, job.first[li] = nb;
            nb += (el + 1023) / 1024;
        }
        job.n = nlayers - 1, job.first[job.n] = nb;
        if (nb > R / 64) job.n = 0;  // (cannot happen for R > 64 * 44; keep the planes off then)
    }
#endif
    // the last layer publishes pool keys (B, 2, Cn) into pool_val (cleared by this call's first kernel) when the FC chain's pool
    // stage follows (pooled == NULL), and when this call finishes the pool itself above 32 clouds (bn_finalize_pool_keys_kernel:
    // the block-partial pick does not scale with the batch) -- provided the caller's (R / 64, 2, Cn) float scratch holds them
    const bool keys_pool = !pooled || (B > 32 && N >= 128 && channels[nlayers] <= 128);
    if (keys_pool) {
        job.zero_keys = reinterpret_cast<unsigned long long *>(pool_val), job.nkeys = B * 2 * channels[nlayers];
        SN_REQUIRE((long long)job.nkeys <= (long long)(R / 64) * 256, "too few rows to clear the pool keys");
    }
    // 64-row blocks per workgroup of the xyz layer: 1 until the batch is large (the headline's 512 blocks stay as they are), then up to
    // 8 while the riders (weight split blocks, key clear: 256 keys per workgroup) still find their workgroups and >= 4 per CU remain
    int in3_nb = 1;
    {
        const long long need = std::max<long long>(job.n > 0 ? job.first[job.n] : 0, keys_pool ? (job.nkeys + 255) / 256 : 0);
        while (in3_nb < 8 && (R / 64) / (in3_nb * 2) >= std::max<long long>(need, 4 * device_cus()) && (R / 64) % (in3_nb * 2) == 0) in3_nb *= 2;
        if (g_in3_blocks > 0 && (R / 64) / g_in3_blocks >= std::max<long long>(need, 1)) in3_nb = g_in3_blocks;  // (test hook)
    }
    hipLaunchKernelGGL(conv_in3_fwd_kernel, dim3((R / 64 + in3_nb - 1) / in3_nb, channels[1] / 64), dim3(256), 0, st, R, channels[1], x, W[0],
                       bias ? bias[0] : nullptr, z[0], (float *)nullptr, acc, acc + (size_t)(nlayers - 1) * S + kFxPoison, job,
                       in3_nb);
    using T = TileBig;
    for (int l = 1; l < nlayers; ++l) {
        const int Ci = channels[l], Co = channels[l + 1];
        FwdArgs g{};
        g.a = make_act(z[l - 1], nullptr, R, Ci);
        g.w.w = W[l], g.w.co = Co, g.w.ci = Ci;
        g.bias = bias ? bias[l] : nullptr, g.z = z[l], g.stats = nullptr;
        g.acc_in = acc + (size_t)(l - 1) * S, g.bn_prev = bn_of(l - 1), g.acc_out = acc + (size_t)l * S;
        if (l >= 2) g.zero_ptr = acc + (size_t)(l - 2) * S, g.zero_n = (int)S;
        if (l == nlayers - 1) {
            g.pool_npts = N;
            if (!keys_pool) g.pool_val = pool_val, g.pool_idx = pool_idx;
            else g.pool_keys = reinterpret_cast<unsigned long long *>(pool_val);  // (decoded by the FC chain's pool stage / below)
        }
        g.wplanes = job.n > 0 ? planes[l] : nullptr;
        const bool pl = g.wplanes != nullptr;
#if SN_BF16X3
        // large batches: the persistent, weight-stationary form (linear_fwd_persist_kernel) once every workgroup has at least
        // g_persist_min_tiles 64-row tiles to walk over
        if (pl && g.z && (l < nlayers - 1 || keys_pool) && (l > 1 || z1free) && (Ci == 64 || Co == 128) && Ci <= 128 && Co <= 128 && R % 64 == 0) {
            const int ntiles = R / 64, per_cu = Co == 128 ? 1 : 2, nmax = device_cus() * per_cu;
            if (g_persist_min_tiles > 0 && ntiles >= g_persist_min_tiles * nmax) {
                const int tpw = (ntiles + nmax - 1) / nmax, nwg = (ntiles + tpw - 1) / tpw;
                if (l == 1) g.x3 = x, g.w3 = W[0], g.b3 = bias ? bias[0] : nullptr;
                int rc = 0;
                // 128 output channels: TWO 256-thread workgroups per CU on 32-row tiles (SN_FWD_HT) instead of one 512-thread
                // workgroup on 64-row tiles -- the eight waves of one tile stage, multiply and store in lockstep (one barrier per
                // tile): a SIMD's two waves are always in the same phase; two independent workgroups drift apart and the
                // dependent MFMA chain of one overlaps the staging / epilogue of the other
                const bool ht = SN_FWD_HT && Co == 128 && l != 1;
                const int nmax_h = device_cus() * 2, tpw_h = 2 * ((ntiles + nmax_h - 1) / nmax_h), nwg_h = (2 * ntiles + tpw_h - 1) / tpw_h;
                using THT = Tile<32, 128, 1, 4>;
                if (l == 1) rc = launch_fwd_persist<T, 64, true>(g, ntiles, tpw, nwg, st);
                else if (ht && Ci == 128) rc = launch_fwd_persist<THT, 128, false>(g, 2 * ntiles, tpw_h, nwg_h, st);
                else if (ht) rc = launch_fwd_persist<THT, 64, false>(g, 2 * ntiles, tpw_h, nwg_h, st);
                else if (Co == 128 && Ci == 128) rc = launch_fwd_persist<SN_FWD_TW, 128, false>(g, ntiles, tpw, nwg, st);
                else if (Co == 128) rc = launch_fwd_persist<SN_FWD_TW, 64, false>(g, ntiles, tpw, nwg, st);
                else rc = launch_fwd_persist<T, 64, false>(g, ntiles, tpw, nwg, st);
                if (rc) return rc;
                continue;
            }
        }
#endif
        if (l == 1 && z1free) {
            SN_REQUIRE(pl, "z[0] == NULL: the weight planes are missing");
            g.x3 = x, g.w3 = W[0], g.b3 = bias ? bias[0] : nullptr;
            const dim3 grid(R / T::BM, Co / T::BN);
            const size_t lds = shaped_lds(lds_bytes<T>() + (size_t)2 * Ci * sizeof(float), grid);
            hipLaunchKernelGGL((linear_fwd_kernel<T, true, ACT_BN_RELU_FX, 64, SN_BF16X3 != 0, SN_BF16X3 != 0>), grid, dim3(T::THREADS), lds, st, g);
            continue;
        }
        if (Co % 128 == 0) {
            // 128 output channels: one 512-thread workgroup per 64 rows computes all of them -- the input tile is fetched
            // once instead of once per 64-column block, and half as many workgroups run the statistics prologue
            // (256 output channels: two such column blocks)
            using TW = SN_FWD_TW;
            const dim3 grid(R / TW::BM, Co / TW::BN);
            const size_t lds = shaped_lds(lds_bytes<TW>() + (size_t)2 * Ci * sizeof(float), grid);
#define SN_FWD_FX(TT, KT_)                                                                                                \
    do {                                                                                                                  \
        if (pl) hipLaunchKernelGGL((linear_fwd_kernel<TT, true, ACT_BN_RELU_FX, KT_, SN_BF16X3 != 0>), grid, dim3(TT::THREADS), lds, st, g); \
        else hipLaunchKernelGGL((linear_fwd_kernel<TT, true, ACT_BN_RELU_FX, KT_>), grid, dim3(TT::THREADS), lds, st, g);  \
    } while (0)
            if (Ci == 256) {  // (sn_conv_stack_forward_supported: wide layers only with the planes)
                SN_REQUIRE(pl, "a 256-channel input needs the pre-split weight planes");
                hipLaunchKernelGGL((linear_fwd_kernel<TW, true, ACT_BN_RELU_FX, 256, SN_BF16X3 != 0>), grid, dim3(TW::THREADS), lds, st, g);
            } else if (Ci == 128 && pl) SN_FWD_FX(TW, 128);
            else if (Ci == 128) SN_FWD_FX(TW, SN_FWD_KT128);
            else SN_FWD_FX(TW, 64);
            continue;
        }
        const dim3 grid(R / T::BM, Co / T::BN);
        const size_t lds = shaped_lds(lds_bytes<T>() + (size_t)2 * Ci * sizeof(float), grid);
        if (Ci == 128 && pl) SN_FWD_FX(T, 128);
        else if (Ci == 128) SN_FWD_FX(T, SN_FWD_KT128);
        else SN_FWD_FX(T, 64);
#undef SN_FWD_FX
    }
    const int Cn = channels[nlayers];
    long long *zp = nlayers >= 2 ? acc + (size_t)(nlayers - 2) * S : nullptr;
    if (!pooled) {  // the last BatchNorm + the pool pick run as the first stage of sn_fc_chain_forward_pool
        SN_LAUNCH_CHECK();
        return 0;
    }
    if (keys_pool) {
        const int cpb = 8;  // clouds per workgroup
        hipLaunchKernelGGL(bn_finalize_pool_keys_kernel, dim3((B + cpb - 1) / cpb), dim3(256), 0, st, bn_of(nlayers - 1), B, Cn, cpb,
                           reinterpret_cast<const unsigned long long *>(pool_val), pooled, argsel, zsel,
                           acc + (size_t)(nlayers - 1) * S, zp, zp ? (int)S : 0);
        SN_LAUNCH_CHECK();
        return 0;
    }
    hipLaunchKernelGGL(bn_finalize_pool_kernel, dim3((Cn + kChan - 1) / kChan), dim3(1024), 0, st, 0, Cn, (const float *)nullptr,
                       bn_of(nlayers - 1), B, N / 64, pool_val, pool_idx, pooled, argsel, zsel, acc + (size_t)(nlayers - 1) * S, zp,
                       zp ? (int)S : 0);
    SN_LAUNCH_CHECK();
    return 0;
}

extern "C" int sn_linear_stats_blocks(int R) { return R > 64 ? (R + TileBig::BM - 1) / TileBig::BM : (R + TileSmall::BM - 1) / TileSmall::BM; }

extern "C" int sn_bn_finalize(int nblk, int C, long long R, const float *stats, const float *gamma, const float *beta,
                              float eps, float momentum, float *running_mean, float *running_var,
                              long long *num_batches_tracked, float *coef, sn_stream_t stream)
{
    SN_REQUIRE(nblk >= 1 && C >= 1 && R >= 1 && stats && gamma && beta && coef, "bad argument");
    const BnFwd bn{gamma, beta, running_mean, running_var, num_batches_tracked, coef, eps, momentum, R};
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + kChan - 1) / kChan), dim3(1024), 0, (hipStream_t)stream, nblk, C, stats, bn);
    SN_LAUNCH_CHECK();
    return 0;
}

// Training-mode BatchNorm of a SHORT activation matrix z (R, C) -- the FC head at batches above 32 -- with two-pass statistics
// (bn_twopass_kernel); outputs as sn_bn_finalize.
extern "C" int sn_bn_batch_stats_twopass(int R, int C, const float *z, const float *gamma, const float *beta, float eps,
                                         float momentum, float *running_mean, float *running_var,
                                         long long *num_batches_tracked, float *coef, sn_stream_t stream)
{
    SN_REQUIRE(R >= 1 && C >= 1 && z && gamma && beta && coef, "bad argument");
    const BnFwd bn{gamma, beta, running_mean, running_var, num_batches_tracked, coef, eps, momentum, (long long)R};
    hipLaunchKernelGGL(bn_twopass_kernel, dim3((C + 63) / 64), dim3(1024), 0, (hipStream_t)stream, R, C, z, bn);
    SN_LAUNCH_CHECK();
    return 0;
}

// The head's OUTPUT layer with a BatchNorm and no activation behind it (classification sampler: classification/models/
// samplenet_model.py:100-108) as ONE launch for R <= 32 rows: z = ain' W^T + bias (ain' = relu(bn(ain)) when coef_prev),
// training-mode batch statistics over the R rows -> coef (4, Co) + running statistics, y = z scale + shift.  Needs
// Ci in {64, 128, 256, 512} (the LDS-staged small-R kernel); otherwise SN_ERR_UNSUPPORTED (compose sn_linear_forward_rows +
// sn_bn_output_forward).
extern "C" int sn_layer_forward_bn_out(int R, int Ci, int Co, const float *ain, const float *coef_prev, const float *W,
                                       const float *bias, float *z, const float *gamma, const float *beta, float eps, float momentum,
                                       float *running_mean, float *running_var, long long *num_batches_tracked, float *coef,
                                       float *y, sn_stream_t stream)
{
    SN_REQUIRE(R >= 1 && Ci >= 1 && Co >= 1, "bad size");
    SN_REQUIRE(ain && W && z && gamma && beta && coef && y, "null pointer");
    if (R > 32 || Ci < 64 || Ci > 512 || (Ci & (Ci - 1)))
        return sn_set_error(SN_ERR_UNSUPPORTED, "sn_layer_forward_bn_out: R <= 32 rows, Ci a power of two in 64 .. 512");
    FwdArgs g{};
    g.a = make_act(ain, coef_prev, R, Ci);
    g.w.w = W, g.w.co = Co, g.w.ci = Ci;
    g.bias = bias, g.z = z, g.stats = nullptr;
    g.bn = BnFwd{gamma, beta, running_mean, running_var, num_batches_tracked, coef, eps, momentum, (long long)R};
    g.bn_y = y;
    hipStream_t st = (hipStream_t)stream;
    if (coef_prev)
        launch_fwd<ACT_BN_RELU>(g, st);
    else
        launch_fwd<ACT_NONE>(g, st);
    SN_LAUNCH_CHECK();
    return 0;
}

// BatchNorm WITHOUT activation on a short matrix z (R, C), C % 4 == 0: training != 0 -- two-pass batch statistics (as
// sn_bn_batch_stats_twopass: coef, running statistics), else coefficients from the running statistics; then y = z scale + shift.
extern "C" int sn_bn_output_forward(int R, int C, int training, const float *z, const float *gamma, const float *beta, float eps,
                                    float momentum, float *running_mean, float *running_var, long long *num_batches_tracked,
                                    float *coef, float *y, sn_stream_t stream)
{
    SN_REQUIRE(R >= 1 && C >= 4 && C % 4 == 0 && z && gamma && beta && coef && y, "bad argument");
    hipStream_t st = (hipStream_t)stream;
    if (training) {
        const BnFwd bn{gamma, beta, running_mean, running_var, num_batches_tracked, coef, eps, momentum, (long long)R};
        hipLaunchKernelGGL(bn_twopass_kernel, dim3((C + 63) / 64), dim3(1024), 0, st, R, C, z, bn);
    } else {
        SN_REQUIRE(running_mean && running_var, "eval mode needs the running statistics");
        hipLaunchKernelGGL(bn_eval_coef_kernel, dim3((C + 63) / 64), dim3(64), 0, st, C, gamma, beta, eps, running_mean, running_var, coef);
    }
    const long long n4 = (long long)R * C / 4;
    hipLaunchKernelGGL(bn_apply_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, n4, C, z, coef, y);
    SN_LAUNCH_CHECK();
    return 0;
}

// Backward of sn_layer_forward_bn_out / sn_bn_output_forward's BatchNorm: see bn_output_bwd_kernel.  fixed != 0: the forward ran
// on running statistics.
extern "C" int sn_bn_output_backward(int R, int C, int fixed, const float *gy, const float *z, const float *coef, float *dz,
                                     float *dgamma, float *dbeta, sn_stream_t stream)
{
    SN_REQUIRE(R >= 1 && C >= 1 && gy && z && coef && dz && dgamma && dbeta, "bad argument");
    hipLaunchKernelGGL(bn_output_bwd_kernel, dim3((C + 63) / 64), dim3(1024), 0, (hipStream_t)stream, R, C, fixed, gy, z, coef, dz,
                       dgamma, dbeta);
    SN_LAUNCH_CHECK();
    return 0;
}

extern "C" int sn_bn_eval_coef(int C, const float *gamma, const float *beta, float eps, const float *running_mean,
                               const float *running_var, float *coef, sn_stream_t stream)
{
    SN_REQUIRE(C >= 1 && gamma && beta && running_mean && running_var && coef, "bad argument");
    hipLaunchKernelGGL(bn_eval_coef_kernel, dim3((C + 63) / 64), dim3(64), 0, (hipStream_t)stream, C, gamma, beta, eps,
                       running_mean, running_var, coef);
    SN_LAUNCH_CHECK();
    return 0;
}

extern "C" int sn_pool_forward(int B, int N, int C, const float *z, const float *coef, float *pooled, int *argsel,
                               float *zsel, sn_stream_t stream)
{
    SN_REQUIRE(B >= 1 && N >= 1 && C >= 1 && z && coef && pooled && argsel && zsel, "bad argument");
    hipLaunchKernelGGL(pool_fwd_kernel, dim3(B, (C + 63) / 64), dim3(1024), 0, (hipStream_t)stream, N, C, z, coef, pooled,
                       argsel, zsel);
    SN_LAUNCH_CHECK();
    return 0;
}
