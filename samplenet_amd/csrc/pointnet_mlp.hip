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
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "sn_common.h"
#include "step_tail.h"

namespace sn {

typedef float f32x16 __attribute__((ext_vector_type(16)));

#ifndef SN_BF16X3
#define SN_BF16X3 1  // conv-stack GEMMs: fp32 products as split-bf16 products on the bf16 matrix cores (gemm_tile_bx3)
#endif
#ifndef SN_FWD_KT128
#define SN_FWD_KT128 0  // 128 input channels: chunk-by-chunk prefetch (fetching the whole K = 128 up front measured slower)
#endif
#ifndef SN_FWD_TW
#define SN_FWD_TW Tile<64, 128, 2, 4>  // conv layers with 128 output channels: one 512-thread workgroup per 64 rows (Tile<64, 128, 2, 2>,
                                       // 32 x 64 per wave, half the LDS fragment traffic: 17.5 vs 15.8 us on the 128 -> 128 layer)
#endif
constexpr int BK = 64;   // K chunk (one chunk covers the 64-channel layers: a single exposed global-load latency)
constexpr int LPAD = 4;  // LDS row padding (floats): keeps rows 16-B aligned for the float4 staging stores

// ------------------------------------------------------------------------------------------------
// Operand loaders.  Each returns 4 consecutive elements along the operand's contiguous dimension,
// already transformed, zero-filled out of bounds.
//   KC (k contiguous):  value(x, k..k+3)     source [X][K]
//   XC (x contiguous):  value(x..x+3, k)     source [K][X]
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float4 ld4_guard(const float *__restrict__ p, size_t off, int valid, bool aligned)
{
    // valid in [0,4]: number of in-bounds elements starting at p[off]
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (valid >= 4 && aligned) return *reinterpret_cast<const float4 *>(p + off);
    if (valid > 0) v.x = p[off];
    if (valid > 1) v.y = p[off + 1];
    if (valid > 2) v.z = p[off + 2];
    if (valid > 3) v.w = p[off + 3];
    return v;
}

enum { ACT_NONE = 0, ACT_BN_RELU = 1, ACT_BN_RELU_FX = 2 };

// Batch statistics as fixed point accumulated with INTEGER atomics: integer addition is associative, so the totals do not
// depend on the order the workgroups arrive in (deterministic, unlike floating-point atomics), and the CONSUMER of the
// statistics (the next layer's GEMM) can finalise the BatchNorm itself from 2 C numbers -- no partials-reduction launch
// between two layers.  A contribution v (an fp32 block sum) is q = v * 2^SHIFT as an integer (exact: v has 24 significant
// bits; sub-resolution tails are rounded to nearest).  |q| < 2^50 -- every realistic value -- goes into the signed "lo"
// accumulator with ONE atomic (<= 2^13 contributions: the sum stays below 2^63).  Larger ones are split q = hi * 2^50 + lo
// (0 <= lo < 2^50) over the lo and a "hi" accumulator (one pair of hi rows per layer, shared by all slots: rare).
// Nothing can overflow for |v| * 2^SHIFT < 2^100; beyond that, or for a non-finite v, the poison word is set
// and the consumer produces NaN coefficients (loud) instead of wrapped sums.  SHIFT = 32 forward (resolution 2.3e-10,
// single-atomic path up to |v| < 2^18), 60 backward (gradient sums down to 1e-18, single-atomic path up to 1e-3).
// The total IS the exact sum of the block sums -- better than the double-precision reduction of partials it replaces.
// Same-address device-scope atomics serialise at the memory side (~20 ns per 128-byte line operation, measured: 512
// workgroups adding into one set of sums cost ~10 us per layer): the workgroups spread over kFxSlots copies
// (swept 4 / 8 / 16 / 32: 243.5 / 240.3 / 239.2 / 255.5 us per step).
#ifndef SN_FX_SLOTS
#define SN_FX_SLOTS 16
#endif
constexpr int kFxSlots = SN_FX_SLOTS;
constexpr int kFxRow = 128;                           // channels per row (C <= 128)
constexpr int kFxHi = kFxSlots * 2 * kFxRow;          // lo rows [slot][stat][128], then ONE pair of hi rows [stat][128]
constexpr int kFxPoison = kFxHi + 2 * kFxRow;         // (large contributions are rare: they all share slot-less hi rows, so
constexpr int kFxLayer = kFxPoison + 64;              //  consumers read them unconditionally -- no flag, no branch)
constexpr int kFxShiftFwd = 32, kFxShiftBwd = 60;
constexpr double kFx2p50 = 1125899906842624.0;
template <int SHIFT>
__device__ __forceinline__ void fx_add(long long *layer, int slot, int stat, int c, float v)
{
    const double d = (double)v * (double)(1ull << 30) * (double)(1ull << (SHIFT - 30));  // exact power-of-two scaling
    long long *lo = layer + (slot * 2 + stat) * kFxRow + c;
    if (fabs(d) < kFx2p50) {
        atomicAdd(reinterpret_cast<unsigned long long *>(lo), (unsigned long long)__double2ll_rn(d));
    } else if (fabs(d) < kFx2p50 * kFx2p50) {
        const double h = floor(d * (1.0 / kFx2p50));
        atomicAdd(reinterpret_cast<unsigned long long *>(lo), (unsigned long long)__double2ll_rn(d - h * kFx2p50));
        atomicAdd(reinterpret_cast<unsigned long long *>(layer + kFxHi + stat * kFxRow + c), (unsigned long long)(long long)h);
    } else {  // (also NaN)
        layer[kFxPoison] = 1;
    }
}
// both totals of channel c (stat 0, stat 1) over the slots, in ONE batch of loads and without a branch: reading the
// statistics one after the other, or behind a "hi rows in use" flag test, costs a second memory round trip per consumer
// (+6 us per step, measured).
// (the loads and the arithmetic separately, for callers that want other fetches issued between the two)
struct FxRaw2 {
    long long lo[kFxSlots][2], hi[2], poison;
};
__device__ __forceinline__ FxRaw2 fx_load2(const long long *layer, int c)
{
    FxRaw2 r;
    r.poison = layer[kFxPoison];
#pragma unroll
    for (int q = 0; q < kFxSlots; ++q) r.lo[q][0] = layer[(q * 2 + 0) * kFxRow + c], r.lo[q][1] = layer[(q * 2 + 1) * kFxRow + c];
    r.hi[0] = layer[kFxHi + c], r.hi[1] = layer[kFxHi + kFxRow + c];
    return r;
}
template <int SHIFT>
__device__ __forceinline__ void fx_total2(const FxRaw2 &r, double &t0, double &t1)
{
    long long a = 0, b = 0;
#pragma unroll
    for (int q = 0; q < kFxSlots; ++q) a += r.lo[q][0], b += r.lo[q][1];
    const double x = (double)a + (double)r.hi[0] * kFx2p50, y = (double)b + (double)r.hi[1] * kFx2p50;
    const double sc = (1.0 / (double)(1ull << 30)) * (1.0 / (double)(1ull << (SHIFT - 30)));
    const double nan = __longlong_as_double(0x7ff8000000000000ll);
    t0 = r.poison ? nan : x * sc;
    t1 = r.poison ? nan : y * sc;
}
template <int SHIFT>
__device__ __forceinline__ void fx_get2(const long long *layer, int c, double &t0, double &t1)
{
    const long long poison = layer[kFxPoison];
    long long a = 0, b = 0;
#pragma unroll
    for (int q = 0; q < kFxSlots; ++q) a += layer[(q * 2 + 0) * kFxRow + c], b += layer[(q * 2 + 1) * kFxRow + c];
    const long long ha = layer[kFxHi + c], hb = layer[kFxHi + kFxRow + c];
    const double x = (double)a + (double)ha * kFx2p50, y = (double)b + (double)hb * kFx2p50;
    const double sc = (1.0 / (double)(1ull << 30)) * (1.0 / (double)(1ull << (SHIFT - 30)));
    const double nan = __longlong_as_double(0x7ff8000000000000ll);
    t0 = poison ? nan : x * sc;
    t1 = poison ? nan : y * sc;
}
// every workgroup clears its share of the accumulators the PREVIOUS kernel consumed (nobody touches them in this launch)
__device__ __forceinline__ void fx_clear_share(long long *p, int n, int block, int nblocks, int tid, int nthreads)
{
    if (!p || tid < 0) return;
    const int per = (n + nblocks - 1) / nblocks;
    const int end = min(n, (block + 1) * per);
    for (int i = block * per + tid; i < end; i += nthreads) p[i] = 0;
}

enum { DZ_PLAIN = 0, DZ_BN = 1, DZ_POOL = 2 };

// ReLU that propagates NaN like torch.relu (fmaxf(NaN, 0) is 0: a diverged run or poisoned statistics would turn into
// plausible-looking zeros instead of a NaN loss)
__device__ __forceinline__ float relu_np(float u) { return u < 0.f ? 0.f : u; }

// Double-precision reciprocal and reciprocal square root from an fp32 hardware seed and two Newton steps (error ~1e-16: the
// result equals the quotient a / x or 1 / sqrt(x) to within one double ulp, i.e. the fp32 coefficients derived from it are
// the same).  The library routines cost ~40 / ~80 double-rate instructions and sat on the critical path of every kernel that
// finalises a BatchNorm in its prologue or epilogue (~0.5 us each, measured on the FC chain).
__device__ __forceinline__ double fast_rcp(double x)
{
    double r = (double)(1.0f / (float)x);
    r = r * (2.0 - x * r);
    return r * (2.0 - x * r);
}
__device__ __forceinline__ double fast_rsqrt(double x)
{
    double y = (double)(1.0f / sqrtf((float)x));
    y = y * (1.5 - 0.5 * x * y * y);
    return y * (1.5 - 0.5 * x * y * y);
}

// activation of the previous layer, rows x channels, channel-contiguous: a = relu(scale[c]*z + shift[c]) or raw
struct ActSrc {
    const float *z;      // [rows][ch]
    const float *scale;  // [ch] (ACT_BN_RELU)
    const float *shift;
    int rows, ch, mode;
    int ones_col;  // if >= 0: channel index that reads as 1.0 (bias column of wgrad)

    // branch-free scalar access (small-R kernels): out-of-range reads are clamped to element 0 and zeroed
    template <int MODE>
    __device__ __forceinline__ float at(int r, int c) const
    {
        const bool ok = r < rows && c < ch;
        const size_t o = ok ? (size_t)r * ch + c : 0;
        float v = z[o];
        if (MODE == ACT_BN_RELU) {
            const int cc = ok ? c : 0;
            v = relu_np(fmaf(v, scale[cc], shift[cc]));
        }
        v *= ok ? 1.f : 0.f;  // mask by multiplication (see small_fwd_kernel): keeps the loads unconditional
        return (ones_col >= 0 && c == ones_col && r < rows) ? 1.f : v;
    }

    // FULL: caller guarantees r < rows, c + 3 < ch, ch % 4 == 0, no ones column: straight 16-byte loads that the
    // compiler can issue back to back (the guarded path puts every load behind its own branch).
    template <bool FULL, int MODE>
    __device__ __forceinline__ float4 load_c4(int r, int c) const  // 4 consecutive channels of row r
    {
        if (FULL) {
            float4 v = *reinterpret_cast<const float4 *>(z + (size_t)r * ch + c);
            if (MODE == ACT_BN_RELU) {
                const float4 s = *reinterpret_cast<const float4 *>(scale + c);
                const float4 t = *reinterpret_cast<const float4 *>(shift + c);
                v.x = relu_np(fmaf(v.x, s.x, t.x));
                v.y = relu_np(fmaf(v.y, s.y, t.y));
                v.z = relu_np(fmaf(v.z, s.z, t.z));
                v.w = relu_np(fmaf(v.w, s.w, t.w));
            }
            return v;
        }
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r >= rows) return v;
        const int valid = min(4, ch - c);
        const bool al = (ch & 3) == 0;
        if (valid > 0) {
            v = ld4_guard(z, (size_t)r * ch + c, valid, al);
            if (MODE == ACT_BN_RELU) {
                const float4 s = ld4_guard(scale, c, valid, al), t = ld4_guard(shift, c, valid, al);
                v.x = relu_np(fmaf(v.x, s.x, t.x));
                v.y = relu_np(fmaf(v.y, s.y, t.y));
                v.z = relu_np(fmaf(v.z, s.z, t.z));
                v.w = relu_np(fmaf(v.w, s.w, t.w));
            }
            if (valid < 2) v.y = 0.f;
            if (valid < 3) v.z = 0.f;
            if (valid < 4) v.w = 0.f;
        }
        if (ones_col >= 0) {
            if (c == ones_col) v.x = 1.f;
            if (c + 1 == ones_col) v.y = 1.f;
            if (c + 2 == ones_col) v.z = 1.f;
            if (c + 3 == ones_col) v.w = 1.f;
        }
        return v;
    }
};

// gradient w.r.t. the pre-BN output of a layer, rows x channels, channel-contiguous
struct DzSrc {
    const float *dy;  // [rows][ch]   (DZ_PLAIN / DZ_BN)
    const float *z;   // [rows][ch]   (DZ_BN / DZ_POOL not needed for the sparse part)
    const float *k1, *k2, *k3;  // [ch]  dz = k1*dy + k2*z + k3
    const float *gsel;          // [B][ch]  (DZ_POOL) gradient at the pooled element
    const int *argsel;          // [B][ch]  (DZ_POOL) row-within-cloud of the pooled element
    int rows, ch, mode, npts;

    template <int MODE>
    __device__ __forceinline__ float at(int r, int c) const  // branch-free scalar access (small-R kernels)
    {
        const bool ok = r < rows && c < ch;
        const int cc = ok ? c : 0;
        const size_t o = ok ? (size_t)r * ch + c : 0;
        float v;
        if (MODE == DZ_PLAIN) {
            v = dy[o];
        } else {
            float d;
            if (MODE == DZ_POOL) {
                const int rr = ok ? r : 0;
                const int b = rr / npts, n = rr - b * npts;
                d = argsel[(size_t)b * ch + cc] == n ? gsel[(size_t)b * ch + cc] : 0.f;
            } else {
                d = dy[o];
            }
            v = fmaf(k1[cc], d, fmaf(k2[cc], z[o], k3[cc]));
        }
        return v * (ok ? 1.f : 0.f);
    }

    template <bool FULL, int MODE>
    __device__ __forceinline__ float4 load_c4(int r, int c) const
    {
        if (FULL) {
            if (MODE == DZ_PLAIN) return *reinterpret_cast<const float4 *>(dy + (size_t)r * ch + c);
            float4 d;
            if (MODE == DZ_POOL) {
                const int b = r / npts, n = r - b * npts;
                const int4 ag = *reinterpret_cast<const int4 *>(argsel + (size_t)b * ch + c);
                const float4 gs = *reinterpret_cast<const float4 *>(gsel + (size_t)b * ch + c);
                d.x = ag.x == n ? gs.x : 0.f;
                d.y = ag.y == n ? gs.y : 0.f;
                d.z = ag.z == n ? gs.z : 0.f;
                d.w = ag.w == n ? gs.w : 0.f;
            } else {
                d = *reinterpret_cast<const float4 *>(dy + (size_t)r * ch + c);
            }
            const float4 zz = *reinterpret_cast<const float4 *>(z + (size_t)r * ch + c);
            const float4 a = *reinterpret_cast<const float4 *>(k1 + c), bb = *reinterpret_cast<const float4 *>(k2 + c),
                         cc = *reinterpret_cast<const float4 *>(k3 + c);
            return make_float4(fmaf(a.x, d.x, fmaf(bb.x, zz.x, cc.x)), fmaf(a.y, d.y, fmaf(bb.y, zz.y, cc.y)),
                               fmaf(a.z, d.z, fmaf(bb.z, zz.z, cc.z)), fmaf(a.w, d.w, fmaf(bb.w, zz.w, cc.w)));
        }
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r >= rows) return v;
        const int valid = min(4, ch - c);
        if (valid <= 0) return v;
        const bool al = (ch & 3) == 0;
        if (MODE == DZ_PLAIN) return ld4_guard(dy, (size_t)r * ch + c, valid, al);
        float4 d;
        if (MODE == DZ_POOL) {
            const int b = r / npts, n = r - b * npts;
            const size_t o = (size_t)b * ch + c;
            d.x = (valid > 0 && argsel[o] == n) ? gsel[o] : 0.f;
            d.y = (valid > 1 && argsel[o + 1] == n) ? gsel[o + 1] : 0.f;
            d.z = (valid > 2 && argsel[o + 2] == n) ? gsel[o + 2] : 0.f;
            d.w = (valid > 3 && argsel[o + 3] == n) ? gsel[o + 3] : 0.f;
        } else {
            d = ld4_guard(dy, (size_t)r * ch + c, valid, al);
        }
        const float4 zz = ld4_guard(z, (size_t)r * ch + c, valid, al);
        const float4 a = ld4_guard(k1, c, valid, al), bb = ld4_guard(k2, c, valid, al), cc = ld4_guard(k3, c, valid, al);
        v.x = fmaf(a.x, d.x, fmaf(bb.x, zz.x, cc.x));
        v.y = valid > 1 ? fmaf(a.y, d.y, fmaf(bb.y, zz.y, cc.y)) : 0.f;
        v.z = valid > 2 ? fmaf(a.z, d.z, fmaf(bb.z, zz.z, cc.z)) : 0.f;
        v.w = valid > 3 ? fmaf(a.w, d.w, fmaf(bb.w, zz.w, cc.w)) : 0.f;
        return v;
    }
};

// weights W [co][ci] row-major
struct WSrc {
    const float *w;
    int co, ci;
    template <bool FULL>
    __device__ __forceinline__ float4 load_ci4(int o, int i) const  // 4 consecutive ci of row co=o
    {
        if (FULL) return *reinterpret_cast<const float4 *>(w + (size_t)o * ci + i);
        if (o >= co) return make_float4(0.f, 0.f, 0.f, 0.f);
        return ld4_guard(w, (size_t)o * ci + i, min(4, ci - i), (ci & 3) == 0);
    }
};

// ------------------------------------------------------------------------------------------------
// GEMM core
// ------------------------------------------------------------------------------------------------
#ifdef SN_TIMELINE
// Debug build only (tools/timeline.sh): per-workgroup phase timestamps (100 MHz wall clock) of the GEMM kernels.
__device__ unsigned long long sn_tl_buf[16384 * 16];
__device__ __forceinline__ void sn_tl(int slot, unsigned long long v)
{
    const unsigned id = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    if (id >= 16384u) return;
    if (threadIdx.x == 0) sn_tl_buf[(size_t)id * 16 + slot] = v;
    if (threadIdx.x == 256) sn_tl_buf[(size_t)id * 16 + 8 + slot] = v;  // a wave of the second kind (fused backward)
}
__device__ unsigned sn_hw_buf[16384 * 8];  // HW_ID of every wave of a workgroup (fused backward)
__device__ __forceinline__ void sn_hw_record()
{
    const unsigned id = blockIdx.x;
    if ((threadIdx.x & 63) == 0 && id < 16384u && (threadIdx.x >> 6) < 8)
        sn_hw_buf[id * 8 + (threadIdx.x >> 6)] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
}
#define SN_TL(slot) sn_tl(slot, wall_clock64())
#define SN_TL_DRAIN() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define SN_TL_ID(kind) sn_tl(7, ((unsigned long long)(kind) << 48) | ((unsigned long long)__builtin_amdgcn_s_getreg((3 << 11) | 20) << 32) | \
                                    (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4))
// the FC chain kernels: [kind 0 forward / 1 backward][workgroup 0..15][stamp 0..31], thread 0 of the workgroup
__device__ unsigned long long sn_fc_tl_buf[2 * 16 * 32];
#define FC_TL(kind, wg, k)                                                              \
    do {                                                                                \
        if (threadIdx.x == 0 && (wg) < 16 && (k) < 32) sn_fc_tl_buf[((kind)*16 + (wg)) * 32 + (k)] = wall_clock64(); \
    } while (0)
#else
#define SN_TL(slot)
#define SN_TL_DRAIN()
#define SN_TL_ID(kind)
#define FC_TL(kind, wg, k)
#endif

template <int BM_, int BN_, int WR_, int WC_>
struct Tile {
    static constexpr int BM = BM_, BN = BN_, WR = WR_, WC = WC_;
    static constexpr int THREADS = WR * WC * 64;
    static constexpr int TM = BM / (WR * 32), TN = BN / (WC * 32);
    // leading dimensions are chosen per staging mode inside gemm_tile (see lds_ld); budget for the larger one
    static constexpr int LDA = BM + LPAD, LDB = BN + LPAD;
    static constexpr int LDS_FLOATS = BK * (LDA + LDB);
    static constexpr int A4 = (BM * BK / 4 + THREADS - 1) / THREADS;  // float4 per thread per chunk
    static constexpr int B4 = (BN * BK / 4 + THREADS - 1) / THREADS;
};

// Stage one operand chunk from registers to LDS (k-major).  KC: v holds 4 consecutive k of one x.
template <int BX, int LD, int N4, int THREADS, bool KC>
__device__ __forceinline__ void stage_store(float *__restrict__ S, const float4 (&v)[N4], int tid)
{
#pragma unroll
    for (int q = 0; q < N4; ++q) {
        const int f = tid + q * THREADS;
        constexpr bool exact = (BX * BK / 4) % THREADS == 0;
        if (exact || f < BX * BK / 4) {
            if (KC) {
                const int x = f / (BK / 4), k4 = (f % (BK / 4)) * 4;
                S[(k4 + 0) * LD + x] = v[q].x;
                S[(k4 + 1) * LD + x] = v[q].y;
                S[(k4 + 2) * LD + x] = v[q].z;
                S[(k4 + 3) * LD + x] = v[q].w;
            } else {
                const int k = f / (BX / 4), x4 = (f % (BX / 4)) * 4;
                *reinterpret_cast<float4 *>(&S[k * LD + x4]) = v[q];
            }
        }
    }
}

// Fetch one K chunk of both operands into registers (float4 per thread, transformed by the loaders).
template <class T, bool A_KC, bool B_KC, class FA, class FB>
__device__ __forceinline__ void fetch_chunk(float4 (&ra)[T::A4], float4 (&rb)[T::B4], const FA &fa, const FB &fb, int k0,
                                            int tid)
{
#pragma unroll
    for (int q = 0; q < T::A4; ++q) {
        const int f = tid + q * T::THREADS;
        // the guard exists only when the tile does not divide evenly among the threads (a branch around a load
        // makes the compiler wait for every load individually)
        constexpr bool exact = (T::BM * BK / 4) % T::THREADS == 0;
        ra[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (exact || f < T::BM * BK / 4) {
            if (A_KC)
                ra[q] = fa(f / (BK / 4), k0 + (f % (BK / 4)) * 4);
            else
                ra[q] = fa((f % (T::BM / 4)) * 4, k0 + f / (T::BM / 4));
        }
    }
#pragma unroll
    for (int q = 0; q < T::B4; ++q) {
        const int f = tid + q * T::THREADS;
        constexpr bool exact = (T::BN * BK / 4) % T::THREADS == 0;
        rb[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (exact || f < T::BN * BK / 4) {
            if (B_KC)
                rb[q] = fb(f / (BK / 4), k0 + (f % (BK / 4)) * 4);
            else
                rb[q] = fb((f % (T::BN / 4)) * 4, k0 + f / (T::BN / 4));
        }
    }
}

// acc[tm][tn] += A(BM x K) . B(K x BN) for this block's tile.  fa(x, k) / fb(x, k) return float4 along the
// operand's contiguous dimension (k for KC, x for XC); x is relative to the tile origin already applied by the caller.
template <class T, bool A_KC, bool B_KC, class FA, class FB>
__device__ __forceinline__ void gemm_tile(f32x16 (&acc)[T::TM][T::TN], int K, const FA &fa, const FB &fb, float *lds)
{
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / T::WC, wc = wave % T::WC;
    // k-contiguous operands are transposed on their way into LDS (4 scalar stores per float4): with a leading
    // dimension = 1 (mod 8) the 64 lanes of a store hit every bank exactly twice (free); +4 keeps the float4 stores
    // of x-contiguous operands 16-byte aligned.  Fragment reads are conflict-free for any leading dimension.
    constexpr int LDA = T::BM + (A_KC ? 1 : LPAD), LDB = T::BN + (B_KC ? 1 : LPAD);
    float *As = lds, *Bs = lds + BK * T::LDA;
    float4 ra[T::A4], rb[T::B4];

    fetch_chunk<T, A_KC, B_KC>(ra, rb, fa, fb, 0, tid);
    for (int k0 = 0; k0 < K; k0 += BK) {
        stage_store<T::BM, LDA, T::A4, T::THREADS, A_KC>(As, ra, tid);
        stage_store<T::BN, LDB, T::B4, T::THREADS, B_KC>(Bs, rb, tid);
        __syncthreads();
        if (k0 == 0) SN_TL(1);
        if (k0 + BK < K) fetch_chunk<T, A_KC, B_KC>(ra, rb, fa, fb, k0 + BK, tid);  // loads in flight under the MFMAs
        const int h = lane >> 5, l31 = lane & 31;
#pragma unroll
        for (int s = 0; s < BK / 2; ++s) {
            float a[T::TM], b[T::TN];
#pragma unroll
            for (int i = 0; i < T::TM; ++i) a[i] = As[(2 * s + h) * LDA + (wr * T::TM + i) * 32 + l31];
#pragma unroll
            for (int j = 0; j < T::TN; ++j) b[j] = Bs[(2 * s + h) * LDB + (wc * T::TN + j) * 32 + l31];
#pragma unroll
            for (int i = 0; i < T::TM; ++i)
#pragma unroll
                for (int j = 0; j < T::TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }
    SN_TL(2);
}

// gemm_tile for the fixed-point statistics chain: the first chunk of both operands was fetched RAW by the caller (in flight
// while it finalised the input's BatchNorm -- two memory round trips overlapped instead of chained), and the A operand is
// transformed (BatchNorm + ReLU: xa(v, k)) on its way from registers to LDS.
template <class T, class FA, class FB, class XA>
__device__ __forceinline__ void gemm_tile_x(f32x16 (&acc)[T::TM][T::TN], int K, const FA &fa, const FB &fb, const XA &xa,
                                            float4 (&ra)[T::A4], float4 (&rb)[T::B4], float *lds)
{
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / T::WC, wc = wave % T::WC;
    constexpr int LDA = T::BM + 1, LDB = T::BN + 1;
    float *As = lds, *Bs = lds + BK * T::LDA;
    for (int k0 = 0; k0 < K; k0 += BK) {
#pragma unroll
        for (int q = 0; q < T::A4; ++q) {
            constexpr bool exact = (T::BM * BK / 4) % T::THREADS == 0;
            const int f = tid + q * T::THREADS;
            if (exact || f < T::BM * BK / 4) ra[q] = xa(ra[q], k0 + (f % (BK / 4)) * 4);
        }
        stage_store<T::BM, LDA, T::A4, T::THREADS, true>(As, ra, tid);
        stage_store<T::BN, LDB, T::B4, T::THREADS, true>(Bs, rb, tid);
        __syncthreads();
        if (k0 == 0) SN_TL(1);
        if (k0 + BK < K) fetch_chunk<T, true, true>(ra, rb, fa, fb, k0 + BK, tid);  // loads in flight under the MFMAs
        const int h = lane >> 5, l31 = lane & 31;
#pragma unroll
        for (int s = 0; s < BK / 2; ++s) {
            float a[T::TM], b[T::TN];
#pragma unroll
            for (int i = 0; i < T::TM; ++i) a[i] = As[(2 * s + h) * LDA + (wr * T::TM + i) * 32 + l31];
#pragma unroll
            for (int j = 0; j < T::TN; ++j) b[j] = Bs[(2 * s + h) * LDB + (wc * T::TN + j) * 32 + l31];
#pragma unroll
            for (int i = 0; i < T::TM; ++i)
#pragma unroll
                for (int j = 0; j < T::TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }
    SN_TL(2);
}

// ------------------------------------------------------------------------------------------------
// fp32 products on the bf16 matrix cores.  Every fp32 operand is split into three bf16 numbers, a = a1 + a2 + a3 (round to
// nearest each time: a1 = bf16(a), a2 = bf16(a - a1), a3 = bf16(a - a1 - a2); 8 significant bits each, the sum is exact to
// the last bit or two of a), and a . b is accumulated in fp32 from the six products a_i b_j with i + j <= 4; the three
// dropped ones are below 2^-23 |a b|.  v_mfma_f32_32x32x16_bf16 multiplies bf16 exactly and accumulates in fp32, and issues
// 16x faster per flop than v_mfma_f32_32x32x2_f32: six of them cover K = 16 in 192 cycles per SIMD, the fp32 MFMA takes 512.
// Measured against fp64 on this layer's shapes the result is as close as the fp32 MFMA's (tools/micro/bf16x3_gemm.hip: mean
// error 1.3-2.0e-8 vs 1.7-1.9e-8 of sum |a b|, max 1.1-1.8e-7 vs 1.3-1.9e-7; nine products change nothing) at 2.5x the rate
// (392 vs 155 fp32-equivalent TFLOP/s with operands in registers).
// Operand fragments of the 32x32x16 form: lane -> row / column (lane & 31), 8 consecutive k at 8 (lane >> 5): a 16-byte
// LDS read per plane from a row-major [x][k] bf16 image -- both operands of the forward GEMM are k-contiguous in memory,
// so the staging is a straight copy (no transposes).  C/D layout as the fp32 form.
// ------------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
constexpr int BKX = 32;       // K chunk of the split-bf16 path
constexpr int LDX = BKX + 8;  // LDS row pitch in bf16: 80 bytes -- 16 consecutive rows' 16-byte fragments tile all 64 banks

__device__ __forceinline__ void split3(float a, __bf16 &h1, __bf16 &h2, __bf16 &h3)
{
    h1 = (__bf16)a;
    const float r1 = a - (float)h1;
    h2 = (__bf16)r1;
    h3 = (__bf16)(r1 - (float)h2);
}
// 4 consecutive k of one row -> the three planes' images (8 bytes each)
template <int BX>
__device__ __forceinline__ void stage_split(__bf16 *__restrict__ P, int x, int k4, const float4 v)
{
    bf16x4 p1, p2, p3;
    const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        __bf16 h1, h2, h3;
        split3(e[t], h1, h2, h3);
        p1[t] = h1, p2[t] = h2, p3[t] = h3;
    }
    *reinterpret_cast<bf16x4 *>(P + (0 * BX + x) * LDX + k4) = p1;
    *reinterpret_cast<bf16x4 *>(P + (1 * BX + x) * LDX + k4) = p2;
    *reinterpret_cast<bf16x4 *>(P + (2 * BX + x) * LDX + k4) = p3;
}
// same with an explicit plane stride and row pitch (elements)
template <int PLANE, int PITCH>
__device__ __forceinline__ void stage_split_p(__bf16 *__restrict__ P, int x, int k4, const float4 v)
{
    bf16x4 p1, p2, p3;
    const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        __bf16 h1, h2, h3;
        split3(e[t], h1, h2, h3);
        p1[t] = h1, p2[t] = h2, p3[t] = h3;
    }
    *reinterpret_cast<bf16x4 *>(P + x * PITCH + k4) = p1;
    *reinterpret_cast<bf16x4 *>(P + PLANE + x * PITCH + k4) = p2;
    *reinterpret_cast<bf16x4 *>(P + 2 * PLANE + x * PITCH + k4) = p3;
}
template <class T>
struct Bx3 {
    static constexpr int A4 = T::BM * BKX / 4 / T::THREADS, B4 = T::BN * BKX / 4 / T::THREADS;
    static_assert(A4 * T::THREADS * 4 == T::BM * BKX && B4 * T::THREADS * 4 == T::BN * BKX, "tile must divide among the threads");
    static constexpr size_t LDS_BYTES = (size_t)(T::BM + T::BN) * 3 * LDX * 2;
};
template <class T, class FA, class FB>
__device__ __forceinline__ void fetch_chunk_x(float4 (&ra)[Bx3<T>::A4], float4 (&rb)[Bx3<T>::B4], const FA &fa, const FB &fb,
                                              int k0, int tid)
{
#pragma unroll
    for (int q = 0; q < Bx3<T>::A4; ++q) {
        const int f = tid + q * T::THREADS;
        ra[q] = fa(f / (BKX / 4), k0 + (f % (BKX / 4)) * 4);
    }
#pragma unroll
    for (int q = 0; q < Bx3<T>::B4; ++q) {
        const int f = tid + q * T::THREADS;
        rb[q] = fb(f / (BKX / 4), k0 + (f % (BKX / 4)) * 4);
    }
}
// one K chunk: registers -> split -> LDS -> barrier -> [prefetch()] -> MFMAs -> barrier
template <class T, class XA, class SB, class PF>
__device__ __forceinline__ void bx3_chunk_g(f32x16 (&acc)[T::TM][T::TN], int k0, const XA &xa, const float4 (&ra)[Bx3<T>::A4],
                                            float *lds, const SB &stage_b, const PF &prefetch);
template <class T, class XA, class PF>
__device__ __forceinline__ void bx3_chunk(f32x16 (&acc)[T::TM][T::TN], int k0, const XA &xa, const float4 (&ra)[Bx3<T>::A4],
                                          const float4 (&rb)[Bx3<T>::B4], float *lds, const PF &prefetch)
{
    bx3_chunk_g<T>(acc, k0, xa, ra, lds, [&](__bf16 *Bp) {
#pragma unroll
        for (int q = 0; q < Bx3<T>::B4; ++q) {
            const int f = threadIdx.x + q * T::THREADS;
            stage_split<T::BN>(Bp, f / (BKX / 4), (f % (BKX / 4)) * 4, rb[q]);
        }
    }, prefetch);
}
template <class T, class XA, class SB, class PF>
__device__ __forceinline__ void bx3_chunk_g(f32x16 (&acc)[T::TM][T::TN], int k0, const XA &xa, const float4 (&ra)[Bx3<T>::A4],
                                            float *lds, const SB &stage_b, const PF &prefetch)
{
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / T::WC, wc = wave % T::WC;
    __bf16 *Ap = reinterpret_cast<__bf16 *>(lds), *Bp = Ap + 3 * T::BM * LDX;
#pragma unroll
    for (int q = 0; q < Bx3<T>::A4; ++q) {
        const int f = tid + q * T::THREADS, k4 = (f % (BKX / 4)) * 4;
        stage_split<T::BM>(Ap, f / (BKX / 4), k4, xa(ra[q], k0 + k4));
    }
    stage_b(Bp);
    __syncthreads();
    prefetch();
#pragma unroll
    for (int kk = 0; kk < BKX / 16; ++kk) {
        bf16x8 a[3][T::TM], b[3][T::TN];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
#pragma unroll
            for (int i = 0; i < T::TM; ++i)
                a[p][i] = *reinterpret_cast<const bf16x8 *>(Ap + (p * T::BM + (wr * T::TM + i) * 32 + l31) * LDX + kk * 16 + 8 * h);
#pragma unroll
            for (int j = 0; j < T::TN; ++j)
                b[p][j] = *reinterpret_cast<const bf16x8 *>(Bp + (p * T::BN + (wc * T::TN + j) * 32 + l31) * LDX + kk * 16 + 8 * h);
        }
        // smallest products first; the tiles of a wave interleaved (independent accumulators back to back)
#define SN_BX3_TERM(PA, PB)                                                                                          \
    _Pragma("unroll") for (int i = 0; i < T::TM; ++i) _Pragma("unroll") for (int j = 0; j < T::TN; ++j) acc[i][j] = \
        __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA][i], b[PB][j], acc[i][j], 0, 0, 0)
        SN_BX3_TERM(0, 2);
        SN_BX3_TERM(2, 0);
        SN_BX3_TERM(1, 1);
        SN_BX3_TERM(0, 1);
        SN_BX3_TERM(1, 0);
        SN_BX3_TERM(0, 0);
#undef SN_BX3_TERM
    }
    __syncthreads();
}
// B operand already split (FwdArgs::wplanes): a thread copies ONE item of 8 consecutive k per plane and chunk -- BN rows x 4 items
template <class T>
struct Bx3P {
    static constexpr int NB = T::BN * (BKX / 8) / T::THREADS;  // items per thread per chunk
    static_assert(NB * T::THREADS == T::BN * (BKX / 8), "tile must divide among the threads");
};
template <class T>
__device__ __forceinline__ void fetch_planes_x(bf16x8 (&rb)[Bx3P<T>::NB][3], const __bf16 *__restrict__ wp, int co, int ci, int col0,
                                               int k0, int tid)
{
#pragma unroll
    for (int q = 0; q < Bx3P<T>::NB; ++q) {
        const int f = tid + q * T::THREADS, x = f / (BKX / 8), k8 = (f % (BKX / 8)) * 8;
#pragma unroll
        for (int p = 0; p < 3; ++p)
            rb[q][p] = *reinterpret_cast<const bf16x8 *>(wp + ((size_t)p * co + col0 + x) * ci + k0 + k8);
    }
}
template <class T>
__device__ __forceinline__ void stage_planes_x(__bf16 *__restrict__ Bp, const bf16x8 (&rb)[Bx3P<T>::NB][3], int tid)
{
#pragma unroll
    for (int q = 0; q < Bx3P<T>::NB; ++q) {
        const int f = tid + q * T::THREADS, x = f / (BKX / 8), k8 = (f % (BKX / 8)) * 8;
#pragma unroll
        for (int p = 0; p < 3; ++p) *reinterpret_cast<bf16x8 *>(Bp + (p * T::BN + x) * LDX + k8) = rb[q][p];
    }
}

// gemm_tile_x on the bf16 matrix cores: acc += xa(A) (BM x K) . B^T (BN x K), both k-contiguous; ra / rb hold the first chunk,
// the next one is fetched under the MFMAs of the current
template <class T, class FA, class FB, class XA>
__device__ __forceinline__ void gemm_tile_bx3(f32x16 (&acc)[T::TM][T::TN], int K, const FA &fa, const FB &fb, const XA &xa,
                                              float4 (&ra)[Bx3<T>::A4], float4 (&rb)[Bx3<T>::B4], float *lds)
{
    for (int k0 = 0; k0 < K; k0 += BKX) {
        float4 na[Bx3<T>::A4], nb[Bx3<T>::B4];
        bx3_chunk<T>(acc, k0, xa, ra, rb, lds, [&] {
            if (k0 + BKX < K) fetch_chunk_x<T>(na, nb, fa, fb, k0 + BKX, threadIdx.x);
        });
        if (k0 + BKX < K) {
#pragma unroll
            for (int q = 0; q < Bx3<T>::A4; ++q) ra[q] = na[q];
#pragma unroll
            for (int q = 0; q < Bx3<T>::B4; ++q) rb[q] = nb[q];
        }
    }
    SN_TL(2);
}
// K known at compile time (NCH chunks): the caller fetched ALL of both operands into registers up front -- one memory round
// trip for the whole tile instead of one per chunk (a chunk's 12 MFMAs per wave are far shorter than a fetch)
template <class T, int NCH, class XA>
__device__ __forceinline__ void gemm_tile_bx3_all(f32x16 (&acc)[T::TM][T::TN], const XA &xa, const float4 (&ra)[NCH][Bx3<T>::A4],
                                                  const float4 (&rb)[NCH][Bx3<T>::B4], float *lds)
{
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) bx3_chunk<T>(acc, ch * BKX, xa, ra[ch], rb[ch], lds, [] {});
    SN_TL(2);
}

// the same two drivers with the B operand copied from pre-split planes (fetch_planes_x)
template <class T, class FA>
__device__ __forceinline__ void fetch_a_x(float4 (&ra)[Bx3<T>::A4], const FA &fa, int k0, int tid)
{
#pragma unroll
    for (int q = 0; q < Bx3<T>::A4; ++q) {
        const int f = tid + q * T::THREADS;
        ra[q] = fa(f / (BKX / 4), k0 + (f % (BKX / 4)) * 4);
    }
}
template <class T, class FA, class XA>
__device__ __forceinline__ void gemm_tile_bx3_p(f32x16 (&acc)[T::TM][T::TN], int K, const FA &fa, const XA &xa, const __bf16 *wp,
                                                int co, int col0, float4 (&ra)[Bx3<T>::A4], bf16x8 (&rp)[Bx3P<T>::NB][3], float *lds)
{
    for (int k0 = 0; k0 < K; k0 += BKX) {
        float4 na[Bx3<T>::A4];
        bf16x8 np[Bx3P<T>::NB][3];
        bx3_chunk_g<T>(acc, k0, xa, ra, lds, [&](__bf16 *Bp) { stage_planes_x<T>(Bp, rp, threadIdx.x); }, [&] {
            if (k0 + BKX < K) {
                fetch_a_x<T>(na, fa, k0 + BKX, threadIdx.x);
                fetch_planes_x<T>(np, wp, co, K, col0, k0 + BKX, threadIdx.x);
            }
        });
        if (k0 + BKX < K) {
#pragma unroll
            for (int q = 0; q < Bx3<T>::A4; ++q) ra[q] = na[q];
#pragma unroll
            for (int q = 0; q < Bx3P<T>::NB; ++q)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) rp[q][pl] = np[q][pl];
        }
    }
    SN_TL(2);
}
// K known at compile time (NCH chunks of 32): the caller fetched the first TWO chunks up front (in flight during the statistics
// prologue); chunk c + 2 is requested into the register set that the staging of chunk c has just consumed -- two chunk periods
// for a fetch to land instead of one MFMA phase (12 MFMAs per wave are far shorter than a fetch).  For K = 64 that is the whole K.
template <class T, int NCH, class FA, class XA>
__device__ __forceinline__ void gemm_tile_bx3_ring_p(f32x16 (&acc)[T::TM][T::TN], const FA &fa, const XA &xa, const __bf16 *wp, int co,
                                                     int col0, float4 (&ra)[2][Bx3<T>::A4], bf16x8 (&rp)[2][Bx3P<T>::NB][3], float *lds)
{
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch)
        bx3_chunk_g<T>(acc, ch * BKX, xa, ra[ch & 1], lds, [&](__bf16 *Bp) { stage_planes_x<T>(Bp, rp[ch & 1], threadIdx.x); }, [&] {
            if (ch + 2 < NCH) {
                fetch_a_x<T>(ra[ch & 1], fa, (ch + 2) * BKX, threadIdx.x);
                fetch_planes_x<T>(rp[ch & 1], wp, co, NCH * BKX, col0, (ch + 2) * BKX, threadIdx.x);
            }
        });
    SN_TL(2);
}

// C/D fragment coordinates of v_mfma_f32_32x32x2_f32: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
__device__ __forceinline__ int frag_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

// Deterministic per-column reduction of two per-lane partials over the block's rows:
// halves of a wave (same column) -> wave rows in index order -> out0/out1[col] (valid for tid < BN).
// FX: out0 / out1 are fixed-point accumulators (long long, see fx_add) that receive the block's sums by integer atomics.
template <class T, bool FX = false>
__device__ __forceinline__ void column_reduce2(float (&p0)[T::TN], float (&p1)[T::TN], float *lds, float *out0,
                                               float *out1, int col0, int ncols)
{
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / T::WC, wc = wave % T::WC;
    float *red = lds;  // [WR][2][BN]
#pragma unroll
    for (int j = 0; j < T::TN; ++j) {
        const float s0 = p0[j] + __shfl_xor(p0[j], 32);
        const float s1 = p1[j] + __shfl_xor(p1[j], 32);
        if (lane < 32) {
            const int c = (wc * T::TN + j) * 32 + lane;
            red[(wr * 2 + 0) * T::BN + c] = s0;
            red[(wr * 2 + 1) * T::BN + c] = s1;
        }
    }
    __syncthreads();
    if (tid < T::BN && col0 + tid < ncols) {
        float a0 = 0.f, a1 = 0.f;
#pragma unroll
        for (int r = 0; r < T::WR; ++r) {
            a0 += red[(r * 2 + 0) * T::BN + tid];
            a1 += red[(r * 2 + 1) * T::BN + tid];
        }
        if (FX) {  // out0 = the layer's accumulator block, out1 unused
            long long *layer = reinterpret_cast<long long *>(out0);
            const int slot = blockIdx.x % kFxSlots;
            fx_add<kFxShiftFwd>(layer, slot, 0, col0 + tid, a0);
            fx_add<kFxShiftFwd>(layer, slot, 1, col0 + tid, a1);
        } else {
            out0[col0 + tid] = a0;
            out1[col0 + tid] = a1;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// forward:  Z[R][Co] = act(Ain)[R][Ci] . W^T + bias ; stats partial [gridDim.x][2][Co]
// ------------------------------------------------------------------------------------------------
// BatchNorm (training) finalisation of one channel from its batch sums: coefficients for the next layer / backward and
// the running-statistics update of torch.nn.BatchNorm1d.  Used by bn_finalize_kernel and, when a workgroup already
// owns all rows of its columns (R <= 32), directly by the forward epilogue (no partials, no extra launch).
struct BnFwd {
    const float *gamma, *beta;
    float *running_mean, *running_var;
    long long *num_batches_tracked;
    float *coef;  // [4][C]: scale, shift, mean, invstd;  NULL: no BatchNorm behind this layer
    float eps, momentum;
    long long R;
};

// (gamma, beta, running mean / var are passed in: the callers fetch them BEFORE their reduction so that the loads overlap it)
struct BnFwdIn {
    float gamma, beta, rmean, rvar;
};
__device__ __forceinline__ BnFwdIn bn_fwd_inputs(const BnFwd &bn, int c)
{
    BnFwdIn in{bn.gamma[c], bn.beta[c], 0.f, 0.f};
    if (bn.running_mean) in.rmean = bn.running_mean[c], in.rvar = bn.running_var[c];
    return in;
}
// (write = false: the value only -- several threads of a workgroup may evaluate the same channel, one of them stores)
__device__ __forceinline__ float2 bn_finalize_channel_mv(const BnFwd &bn, int C, int c, double mean, double var, const BnFwdIn &in,
                                                         bool write = true);
__device__ __forceinline__ float2 bn_finalize_channel(const BnFwd &bn, int C, int c, double s, double ss, const BnFwdIn &in,
                                                      bool write = true)
{
    const double rR = fast_rcp((double)bn.R);
    const double mean = s * rR;
    double var = ss * rR - mean * mean;
    if (var < 0.0) var = 0.0;
    return bn_finalize_channel_mv(bn, C, c, mean, var, in, write);
}
__device__ __forceinline__ float2 bn_finalize_channel_mv(const BnFwd &bn, int C, int c, double mean, double var, const BnFwdIn &in,
                                                         bool write)
{
    const float invstd = (float)fast_rsqrt(var + (double)bn.eps);
    const float sc = in.gamma * invstd;
    if (!write) return make_float2(sc, in.beta - (float)mean * sc);
    bn.coef[c] = sc;
    bn.coef[C + c] = in.beta - (float)mean * sc;
    bn.coef[2 * C + c] = (float)mean;
    bn.coef[3 * C + c] = invstd;
    if (bn.running_mean) {
        const double unbiased = bn.R > 1 ? var * (double)bn.R * fast_rcp((double)(bn.R - 1)) : var;
        bn.running_mean[c] = (1.f - bn.momentum) * in.rmean + bn.momentum * (float)mean;
        bn.running_var[c] = (1.f - bn.momentum) * in.rvar + bn.momentum * (float)unbiased;
    }
    return make_float2(sc, in.beta - (float)mean * sc);  // (scale, shift)
}
__device__ __forceinline__ void bn_finalize_channel(const BnFwd &bn, int C, int c, double s, double ss)
{
    bn_finalize_channel(bn, C, c, s, ss, bn_fwd_inputs(bn, c));
}

// BatchNorm backward coefficients of one channel from (sum dY, sum dY*Z):  dZ = k1 dY + k2 Z + k3
struct BnBwd {
    const float *coef;  // [4][C] of that layer;  NULL: nothing to do
    float *dgamma, *dbeta, *dbias, *kcoef;
    long long R;
};

struct BnBwdIn {
    float scale, mean, invstd;
};
__device__ __forceinline__ BnBwdIn bn_bwd_inputs(const BnBwd &bb, int C, int c)
{
    return BnBwdIn{bb.coef[c], bb.coef[2 * C + c], bb.coef[3 * C + c]};
}
struct BnBwdOut {
    float k1, k2, k3, dgamma, dbeta, dbias;
};
__device__ __forceinline__ BnBwdOut bn_backward_coefs(long long R, double s, double sz, const BnBwdIn &in)
{
    const double scale = in.scale, mean = in.mean, invstd = in.invstd;
    const double dg = invstd * (sz - mean * s);
    // R <= 0: the forward normalised with FIXED statistics (eval mode, running mean / variance): dZ = scale * dY, no
    // dependence of the statistics on Z -> k2 = k3 = 0; dgamma / dbeta keep their form (mean, invstd = the fixed ones)
    const double rinv = R > 0 ? fast_rcp((double)R) : 0.0;
    BnBwdOut o;
    o.dgamma = (float)dg, o.dbeta = (float)s;
    o.k1 = (float)scale, o.k2 = (float)(-scale * invstd * dg * rinv);
    o.k3 = (float)(scale * (invstd * mean * dg * rinv - s * rinv));
    o.dbias = (float)((double)o.k1 * s + (double)o.k2 * (double)R * mean + (double)R * (double)o.k3);
    return o;
}
__device__ __forceinline__ float3 bn_backward_channel(const BnBwd &bb, int C, int c, double s, double sz, const BnBwdIn &in)
{
    const BnBwdOut o = bn_backward_coefs(bb.R, s, sz, in);
    bb.dgamma[c] = o.dgamma;
    bb.dbeta[c] = o.dbeta;
    bb.kcoef[c] = o.k1, bb.kcoef[C + c] = o.k2, bb.kcoef[2 * C + c] = o.k3;
    if (bb.dbias) bb.dbias[c] = o.dbias;
    return make_float3(o.k1, o.k2, o.k3);
}
__device__ __forceinline__ void bn_backward_channel(const BnBwd &bb, int C, int c, double s, double sz)
{
    bn_backward_channel(bb, C, c, s, sz, bn_bwd_inputs(bb, C, c));
}

struct FwdArgs {
    ActSrc a;
    WSrc w;
    const float *bias;
    float *z;
    float *stats;  // may be null
    BnFwd bn;      // small-R kernels only: finalise the BatchNorm in the epilogue (bn.coef != NULL)
    // last conv layer (FULL tiles, 64-row blocks inside one cloud): per block and column the maximum and minimum of the
    // pre-BN output with their first row -- the max-pool over the points is then finished by bn_finalize_pool_kernel
    float *pool_val;  // [gridDim.x][2][Co]  (max, min)
    int *pool_idx;    // [gridDim.x][2][Co]  row index inside the cloud
    int pool_npts;
    // fixed-point statistics chain (ACT_BN_RELU_FX, sn_conv_stack_forward_bn): the input's BatchNorm is finalised HERE from
    // acc_in [2][Ci] (every workgroup computes the Ci coefficient pairs into LDS; workgroup (0,0) also stores coef_prev and
    // updates the running statistics), this layer's sums go to acc_out [2][Co] by integer atomics, and zero_ptr [zero_n]
    // (the accumulators the PREVIOUS kernel consumed: nobody touches them during this launch) is cleared for the next step.
    const long long *acc_in;
    BnFwd bn_prev;
    long long *acc_out;
    long long *zero_ptr;
    int zero_n;
    // the weights already split into three bf16 planes [3][Co][Ci] (by the xyz-layer kernel of the same stack call, once per step):
    // staged as straight copies.  NULL: every workgroup splits its W tile itself.
    const __bf16 *wplanes;
    // IN3A (second layer of the stack): the input activation is not read from memory but rebuilt from the cloud -- Z1[r][c] =
    // (W3[c] . x_r) + b3[c], the xyz layer's own expression (conv_in3_fwd_kernel), 3 FMAs per element instead of a 4-byte load:
    // the stack never writes its first activation tensor.  x3 (R, 3), w3 (Ci, 3), b3 (Ci) or NULL.
    const float *x3, *w3, *b3;
    // last conv layer in front of the FC chain's pool stage: instead of block partials (pool_val / pool_idx) the epilogue
    // publishes, per cloud and channel, the maximum and the minimum of Z with its first row as 64-bit keys combined by atomicMax
    // (order-independent; pool_keys [B][2][Co], zero before the launch): the consumer picks by the sign of the BatchNorm scale.
    unsigned long long *pool_keys;
    int pool_max_only;  // the consumer's scale is known to be >= 0 (plain ReLU): the minima are not published
};
// (value, row) -> key: larger value first, then the LOWER row; value order via the usual sign flip of the float bits
__device__ __forceinline__ unsigned long long pool_key(float v, int row)
{
    const unsigned u = __float_as_uint(v);
    const unsigned o = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    return ((unsigned long long)o << 32) | (unsigned)(0xFFFFFFFFu - (unsigned)row);
}
__device__ __forceinline__ void pool_key_decode(unsigned long long k, float &v, int &row)
{
    const unsigned o = (unsigned)(k >> 32);
    const unsigned u = (o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o;
    v = __uint_as_float(u);
    row = (int)(0xFFFFFFFFu - (unsigned)k);
}

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
        const int c = threadIdx.x;  // THREADS >= Ci (<= 128)
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
        column_reduce2<T, true>(s0, s1, lds, reinterpret_cast<float *>(g.acc_out), nullptr, col0, Co);
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
                const int f = tid + q * T::THREADS, x = f / (BKX / 4), k = ch * BKX + (f % (BKX / 4)) * 4;
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
                stage_split<T::BM>(Ap, f / (BKX / 4), k4, xa(ra[ch][q], ch * BKX + k4));
            }
        }
    };
    // running per-column sums (threads tid < BN) and the current cloud's pool keys (lanes < 32 of every wave)
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
        if (tile + 1 < tile1) stage(nxt, buf ^ 1);
        if (tile + 3 < tile1) fetch_tile(nxt, tile + 3);
        const int row0 = tile * T::BM;
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
        const __bf16 *At = Abuf + buf * ABUF;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const __bf16 *Ap = At + (ks / (BKX / 16)) * ACH;
            const int kk = ks % (BKX / 16);
            bf16x8 a[3];
#pragma unroll
            for (int p = 0; p < 3; ++p) a[p] = *reinterpret_cast<const bf16x8 *>(Ap + (p * T::BM + wr * 32 + l31) * LDX + kk * 16 + 8 * h);
            // smallest products first (the order of bx3_chunk_g)
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], breg[ks][2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], breg[ks][0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], breg[ks][1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], breg[ks][1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], breg[ks][0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], breg[ks][0], acc, 0, 0, 0);
        }
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
            if (g.z) {
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
            fx_local_add<kFxShiftFwd>(fx0, g.acc_out, 0, tid, a0);
            fx_local_add<kFxShiftFwd>(fx1, g.acc_out, 1, tid, a1);
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

template <int CI, int CO>
struct CbxShape {
    static constexpr int TR = CO == 128 ? 32 : 64;                  // two tile buffers of three planes must fit the LDS
    static constexpr int KS = (CI == 64 && CO == 128) ? 2 : 1;      // 32 x 64 dgrad block = two 32 x 32 tiles: four waves split K too
    static constexpr int LDZ = CO + 8, LDP = CI + 8;                // bf16 pitches
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
template <int CO, int CI, int TR, int ZMODE, bool FULLR, int NZ4, int NP4, bool RZ1 = false>
__device__ __forceinline__ void cbx_stage(const ConvBwdArgs &g, int tile, int n0, int tid, __bf16 *__restrict__ Zb,
                                          __bf16 *__restrict__ Pb, const float4 (&rz)[NZ4], const float4 (&rdy)[NZ4],
                                          const float4 (&rp)[NP4], const int4 &rag, const float4 &rgs, const float4 &k1,
                                          const float4 &k2, const float4 &k3, const float4 &sc4, const float4 &sh4,
                                          const float4 *w3 = nullptr)
{
    constexpr int ZSTEP = 256 / (CO / 4), PSTEP = 256 / (CI / 4);
    constexpr int LDZ = CO + 8, LDP = CI + 8;
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
        stage_split_p<TR * LDZ, LDZ>(Zb, rt, zc4, v);
    }
#pragma unroll
    for (int q = 0; q < NP4; ++q) {
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
template <int CI, int CO, int ZMODE, bool FULLR, bool IN3 = false, bool RZ1 = false, int GZ = CO, int GP = CI, int GW = CI, int DM = 0>
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
        const int bstep = G / tpc, tstep = G - bstep * tpc;
        int cloud = (int)blockIdx.x / tpc, tic = (int)blockIdx.x - cloud * tpc;
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

        int tile = blockIdx.x;
        cbf_issue_loads<CO, CI, TR, ZMODE, NZ4, NP4, RZ1, GZ, GP>(rs, tile, cloud, zvo, pvo, avo, rz, rdy, rp, rag, rgs);
        if (RZ1) load_xyz_rows(tile);
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
        cbx_stage<CO, CI, TR, ZMODE, FULLR, NZ4, NP4, RZ1>(g, tile, tic * TR, tid, Lb, Lb + 3 * ZPL, rz, rdy, rp, rag, rgs, k1, k2, k3, sc4, sh4,
                                                           w3s);
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
        for (int it = 0; tile < g.ntiles; ++it, tile += G) {
            const __bf16 *Zb = Lb + (it & 1) * BUF;
            if (!IN3 && it > 0 && kh == 0) {
                const unsigned oso = (unsigned)(tile - G) * (TR * GP * 4);
#pragma unroll
                for (int i = 0; i < 4; ++i) buf_store4(vout[i], rs.dyprev, ovo, oso + i * (8 * GP * 4));
            }
            float zq[16], pq[16];
            if (!RZ1 && DM != 1 && (KS == 1 || kh == 0))
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    zq[e] = buf_load1(rs.zprev, qvo, (unsigned)tile * (TR * GP * 4) + ((e & 3) + 8 * (e >> 2)) * (GP * 4));
            if (DM == 2)  // the first pass's raw data gradient at this lane's fragment positions
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    pq[e] = buf_load1(rs.dyacc, qvo, (unsigned)tile * (TR * GP * 4) + ((e & 3) + 8 * (e >> 2)) * (GP * 4));
            const bool more = tile + G < g.ntiles;
            const int nxt = more ? tile + G : tile;
            int ncloud = cloud, ntic = tic;
            if (more) {
                ncloud += bstep, ntic += tstep;
                if (ntic >= tpc) ntic -= tpc, ++ncloud;
            }
            cbf_issue_loads<CO, CI, TR, ZMODE, NZ4, NP4, RZ1, GZ, GP>(rs, nxt, ncloud, zvo, pvo, avo, rz, rdy, rp, rag, rgs);
            if (RZ1) load_xyz_rows(nxt);
            if (xthr) rx = buf_load4(rsx, (unsigned)tid * 16, (unsigned)nxt * (TR * 12));
            // the requests go out HERE: left alone, the scheduler sinks them below the MFMAs to their first use (the staging),
            // and every tile pays a full memory round trip
            __builtin_amdgcn_sched_barrier(0);
            if (it == 1) SN_TL(5);

            f32x16 acc;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = 0.f;
            const __bf16 *ap = Zb + (rb * 32 + l31) * LDZ + kh * KD * 16 + 8 * h;
#pragma unroll
            for (int kk = 0; kk < KD; ++kk) {
                const bf16x8 a0 = *reinterpret_cast<const bf16x8 *>(ap + kk * 16);
                const bf16x8 a1 = *reinterpret_cast<const bf16x8 *>(ap + ZPL + kk * 16);
                const bf16x8 a2 = *reinterpret_cast<const bf16x8 *>(ap + 2 * ZPL + kk * 16);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, wf[kk][2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, wf[kk][0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, wf[kk][1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, wf[kk][1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, wf[kk][0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, wf[kk][0], acc, 0, 0, 0);
            }
            if (it == 1) SN_TL(1);
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
            if (kh == 0) {
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
            if (more) {
                __bf16 *Zn = Lb + ((it + 1) & 1) * BUF;
                cbx_stage<CO, CI, TR, ZMODE, FULLR, NZ4, NP4, RZ1>(g, nxt, ntic * TR, tid, Zn, Zn + 3 * ZPL, rz, rdy, rp, rag, rgs, k1, k2, k3,
                                                                   sc4, sh4, w3s);
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
        if (!IN3 && tile != (int)blockIdx.x && kh == 0) {
            const unsigned oso = (unsigned)(tile - G) * (TR * GP * 4);
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
        __syncthreads();
        // transposing reads: this lane's row / channel offsets inside a [16 rows][32 channels] fragment block
        const int trr = 8 * h + ((lane & 15) >> 2), trc = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
        int tile = blockIdx.x;
        for (int it = 0; tile < g.ntiles; ++it, tile += G) {
            const __bf16 *Zb = Lb + (it & 1) * BUF, *Pb = Zb + 3 * ZPL;
            const __bf16 *ap = Zb + trr * LDZ + cob * 32 + trc;
            const __bf16 *bp = Pb + trr * LDP + trc;
#pragma unroll
            for (int kk = 0; kk < KW; ++kk) {
                bf16x8 a[3], b[3][NWT];
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    a[p] = lds_tr8(ap + p * ZPL + kk * 16 * LDZ, LDZ);
#pragma unroll
                    for (int n = 0; n < NWT; ++n) b[p][n] = lds_tr8(bp + p * PPL + kk * 16 * LDP + ((q0 + n) % NCB) * 32, LDP);
                }
#define SN_BX3_TERM(PA, PB) \
    _Pragma("unroll") for (int n = 0; n < NWT; ++n) accw[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA], b[PB][n], accw[n], 0, 0, 0)
                SN_BX3_TERM(0, 2);
                SN_BX3_TERM(2, 0);
                SN_BX3_TERM(1, 1);
                SN_BX3_TERM(0, 1);
                SN_BX3_TERM(1, 0);
                SN_BX3_TERM(0, 0);
#undef SN_BX3_TERM
            }
            if (it == 1) SN_TL(1);
            if (KS == 2) __syncthreads();  // (the dgrad waves' partial-tile hand-off)
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

// ------------------------------------------------------------------------------------------------
// Small-R kernels (R <= 32: the FC head at the reference batch size).  One 32-row MFMA tile; the GEMM is
// latency-bound, so there is no LDS staging loop: every lane loads its whole K slice of both operands straight
// into registers (all loads in flight at once), the four waves of a workgroup split K, and their accumulators
// are summed through LDS in wave order (deterministic).  MFMA step t of a lane consumes k = kbase + 32*half + t:
// any permutation of k is valid as long as A and B use the same one.
// ------------------------------------------------------------------------------------------------
constexpr int KP = 32;  // k values per lane per pass

__device__ __forceinline__ void wave_sum_to_wave0(f32x16 &acc, float *lds)
{
    // lds: [3][16][64] floats.  waves 1..3 publish, wave 0 adds them in order.
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wave > 0)
#pragma unroll
        for (int e = 0; e < 16; ++e) lds[((wave - 1) * 16 + e) * 64 + lane] = acc[e];
    __syncthreads();
    if (wave == 0)
#pragma unroll
        for (int w = 0; w < 3; ++w)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] += lds[(w * 16 + e) * 64 + lane];
}

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
            av[q] = *reinterpret_cast<const float4 *>(g.a.z + (size_t)min(r, R - 1) * Ci + c4);
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
            const float ma = r < R ? 1.f : 0.f, mw = col0 + r < Co ? 1.f : 0.f;
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
        if (row < R && colok) {
            s0 += v;
            s1 += v * v;
            if (!vec_out) g.z[(size_t)row * Co + col] = v;
        }
        Ts[row * 36 + l31] = v;
    }
    if (vec_out) {  // transposed through LDS: 4 x 16-byte stores per lane instead of 16 dword stores
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = 8 * i + (lane >> 3);
            const float4 v = *reinterpret_cast<const float4 *>(Ts + row * 36 + (lane & 7) * 4);
            if (row < R) *reinterpret_cast<float4 *>(g.z + (size_t)row * Co + col0 + (lane & 7) * 4) = v;
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
struct FcChainArgs {
    const float *a0;  // (R, C0): input of the first layer, used as is (pooled features)
    int R, C0, H, nl;
    FcChainPool P;
    FcChainLayer L[kFcChainMaxLayers];
    float *xbuf;     // [2][32][H] exchange slabs
    unsigned *sync;  // [0] epoch, [1 + s] arrivals at seam s, [15] error flag -- persistent, zero-initialised once
    double rinv_rows, unbias;  // 1 / R and R / (R - 1) (1 when R == 1)
};

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains the wave's GLOBAL loads and stores
// (s_waitcnt vmcnt(0) in front of s_barrier): every barrier of the chain kernels then exposed the full latency of the operand
// prefetches in flight across it (timestamps: ~1.7 us per "MFMA phase" that holds 0.4 us of MFMAs).  Global data never
// crosses these barriers (hand-offs are drained explicitly before their arrival atomics).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ void wave_sum_to_wave0_lds(f32x16 &acc, float *lds)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wave > 0)
#pragma unroll
        for (int e = 0; e < 16; ++e) lds[((wave - 1) * 16 + e) * 64 + lane] = acc[e];
    lds_barrier();
    if (wave == 0)
#pragma unroll
        for (int w = 0; w < 3; ++w)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] += lds[(w * 16 + e) * 64 + lane];
}

// Cross-wave sum of the four K partials of a 32 x 32 tile with the RESULT SPREAD OVER THE FOUR WAVES (a reduce-scatter): every
// wave publishes its 16 accumulator values per lane (4 x 16-byte LDS stores), then wave w sums the partials of columns
// 8 w .. 8 w + 7 -- lane -> (column 8 w + (lane >> 3), rows 4 (lane & 7) .. + 3: four consecutive rows are four consecutive
// accumulator registers of one source lane, i.e. one 16-byte read per partial) -- in wave order ((p0 + p1) + p2) + p3, the order
// in which wave 0 used to add them alone.  The epilogue behind it (bias, BatchNorm statistics over the column's 32 rows = the 8
// lanes of a column: three DPP steps, coefficients, activation) then runs on all four waves instead of one.
constexpr int kRsPitch = 20;  // floats per lane in the exchange (16 + pad: 80-byte stride)
constexpr int kRsFloats = 4 * 64 * kRsPitch;
__device__ __forceinline__ float4 wave_reduce_scatter4(const f32x16 &acc, float *lds)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float *mine = lds + ((size_t)wave * 64 + lane) * kRsPitch;
#pragma unroll
    for (int q = 0; q < 4; ++q) *reinterpret_cast<float4 *>(mine + 4 * q) = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
    lds_barrier();
    const int rg = lane & 7, src = (wave * 8 + (lane >> 3)) + 32 * (rg & 1);  // source lane: column + 32 * (row half)
    const float *p = lds + (size_t)src * kRsPitch + 4 * (rg >> 1);
    float4 v = *reinterpret_cast<const float4 *>(p);
#pragma unroll
    for (int w = 1; w < 4; ++w) {
        const float4 o = *reinterpret_cast<const float4 *>(p + (size_t)w * 64 * kRsPitch);
        v.x += o.x, v.y += o.y, v.z += o.z, v.w += o.w;
    }
    return v;
}
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

template <int C0T, int NLT, bool POOL = false>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) fc_chain_fwd_kernel(FcChainArgs g)
{
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
            if (l == nl - 1 && s_bad) zv.x = zv.y = zv.z = zv.w = __builtin_nanf("");
            *reinterpret_cast<float4 *>(Lr.z + (size_t)trow * H + col0 + tc4) = zv;
        }
        if (l == nl - 1) break;
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
        lds_barrier();
        FC_TL(0, wg, 7 + 6 * l);
    }
    SN_TL_DRAIN();
    FC_TL(0, wg, 31);
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
                                                           long long *__restrict__ clear_flags = nullptr, WSplitJob job = WSplitJob{})
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
    __shared__ float red[2][16][64];
    const int q = threadIdx.x & 15, rs = threadIdx.x >> 4;
    const int cl = q * 4, co = blockIdx.y * 64 + cl;
    const int row0 = blockIdx.x * 64;
    const bool vec = (Co & 3) == 0 && co + 3 < Co;
    float w[4][3], b[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = min(co + j, Co - 1);
        const float m = co + j < Co ? 1.f : 0.f;
        w[j][0] = W[c * 3 + 0] * m, w[j][1] = W[c * 3 + 1] * m, w[j][2] = W[c * 3 + 2] * m;
        b[j] = bias ? bias[c] * m : 0.f;
    }
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
                fx_add<kFxShiftFwd>(acc_out, blockIdx.x % kFxSlots, 0, c, a0);
                fx_add<kFxShiftFwd>(acc_out, blockIdx.x % kFxSlots, 1, c, a1);
            } else {
                float *st = stats + (size_t)blockIdx.x * 2 * Co;
                st[c] = a0;
                st[Co + c] = a1;
            }
        }
    }
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

// ------------------------------------------------------------------------------------------------
// BatchNorm bookkeeping (tiny kernels, one thread per channel)
// ------------------------------------------------------------------------------------------------
// Sum the [nblk][2][C] partials of channel c over the workgroup's 128 slices (8 channels x 128 slices = 1024 threads:
// with the usual 256..512 row blocks every thread has at most 8 loads, all in flight together -- one memory round trip),
// in double, fixed order.  Returns true on the threads (slice 0) that hold the totals.
constexpr int kSlices = 128, kChan = 8;
typedef double PsRed[kSlices][kChan];
typedef double PsRed2[16][kChan];
__device__ __forceinline__ bool partial_sums_in(PsRed *red, PsRed2 *red2, int nblk, int C, const float *__restrict__ stats,
                                                int cblock, double &s0, double &s1, int bstride)
{
    const size_t bs = bstride > 0 ? (size_t)bstride : (size_t)2 * C;  // floats between the partials of consecutive blocks
    const int cl = threadIdx.x & (kChan - 1), sl = threadIdx.x >> 3;
    const int c = cblock * kChan + cl;
    double a0 = 0.0, a1 = 0.0;
    if (c < C) {
        int b = sl;
        for (; b + 3 * kSlices < nblk; b += 4 * kSlices) {  // 8 independent loads in flight
            const float x0 = stats[(size_t)b * bs + c], y0 = stats[(size_t)b * bs + C + c];
            const float x1 = stats[(size_t)(b + kSlices) * bs + c], y1 = stats[(size_t)(b + kSlices) * bs + C + c];
            const float x2 = stats[(size_t)(b + 2 * kSlices) * bs + c], y2 = stats[(size_t)(b + 2 * kSlices) * bs + C + c];
            const float x3 = stats[(size_t)(b + 3 * kSlices) * bs + c], y3 = stats[(size_t)(b + 3 * kSlices) * bs + C + c];
            a0 += ((double)x0 + (double)x1) + ((double)x2 + (double)x3);
            a1 += ((double)y0 + (double)y1) + ((double)y2 + (double)y3);
        }
        for (; b < nblk; b += kSlices) {
            a0 += (double)stats[(size_t)b * bs + c];
            a1 += (double)stats[(size_t)b * bs + C + c];
        }
    }
    red[0][sl][cl] = a0, red[1][sl][cl] = a1;
    __syncthreads();
    if (sl < 16) {  // slices 8 sl .. 8 sl + 7
        double t0 = 0.0, t1 = 0.0;
#pragma unroll
        for (int q = 0; q < 8; ++q) t0 += red[0][sl * 8 + q][cl], t1 += red[1][sl * 8 + q][cl];
        red2[0][sl][cl] = t0, red2[1][sl][cl] = t1;
    }
    __syncthreads();
    if (sl != 0 || c >= C) return false;
    s0 = 0.0, s1 = 0.0;
#pragma unroll
    for (int q = 0; q < 16; ++q) s0 += red2[0][q][cl], s1 += red2[1][q][cl];
    return true;
}

__device__ __forceinline__ bool partial_sums(int nblk, int C, const float *__restrict__ stats, int cblock, double &s0, double &s1)
{
    __shared__ double red[2][kSlices][kChan];
    __shared__ double red2[2][16][kChan];
    return partial_sums_in(red, red2, nblk, C, stats, cblock, s0, s1, 0);
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
__global__ void __launch_bounds__(1024) post_bwd_kernel(int nred, int nsplit, int Co, int Ci, const float *__restrict__ part,
                                                        float *__restrict__ dW, int nblk, int C,
                                                        const float *__restrict__ stats, BnBwd bb)
{
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

// ------------------------------------------------------------------------------------------------
// host-side launch helpers
// ------------------------------------------------------------------------------------------------
using TileBig = Tile<64, 64, 2, 2>;     // R large: 64 rows x 64 cols per 256-thread workgroup -> >= 2 workgroups per CU
                                        // at B*N = 32768 rows, so one workgroup's load latency hides under another's MFMAs
using TileSmall = Tile<32, 128, 1, 4>;  // R small (FC head at small batch): 32 rows x 128 cols
using TileW = Tile<64, 64, 2, 2>;       // weight gradient: Co x (Ci+1) output tile

template <class T>
static size_t lds_bytes()
{
    return sizeof(float) * std::max(T::LDS_FLOATS, T::WR * 2 * T::BN);
}

}  // namespace sn

using namespace sn;

static ActSrc make_act(const float *z, const float *coef, int rows, int ch, int ones_col = -1)
{
    ActSrc a{};
    a.z = z, a.rows = rows, a.ch = ch, a.ones_col = ones_col;
    a.mode = coef ? ACT_BN_RELU : ACT_NONE;
    a.scale = coef, a.shift = coef ? coef + ch : nullptr;
    return a;
}

// Occupancy shaping.  The dispatcher stacks workgroups on a CU up to its resource limit before moving on, so a grid of
// 512 small workgroups can land 4-deep on half of the 256 CUs (measured: SQ_WAIT_INST_ANY 52 % of wave cycles -- four
// waves per SIMD queueing on one matrix pipe) instead of 2-deep on all of them.  Requesting 160 KB / (workgroups per CU
// the grid needs) of LDS makes exactly that many fit, which spreads the grid evenly.
static size_t shaped_lds(size_t needed, dim3 grid)
{
    const size_t nblk = (size_t)grid.x * grid.y * grid.z;
    const size_t per_cu = std::max<size_t>(1, (nblk + 255) / 256);
    const size_t want = std::min<size_t>(64 * 1024, (160 * 1024) / per_cu);
    return std::max(needed, want > 1024 ? want - 1024 : needed);
}

// ---- dispatch helpers: tile x fast-path x operand modes are template parameters (no control flow around loads) ----
#define SN_LAUNCH_T(KERN, T_, FULL_, GRID, ARGS, ...)                                                              \
    do {                                                                                                           \
        const size_t lds_ = shaped_lds(lds_bytes<T_>(), GRID);                                                     \
        if (FULL_)                                                                                                 \
            hipLaunchKernelGGL((KERN<T_, true, __VA_ARGS__>), GRID, dim3(T_::THREADS), lds_, st, ARGS);            \
        else                                                                                                       \
            hipLaunchKernelGGL((KERN<T_, false, __VA_ARGS__>), GRID, dim3(T_::THREADS), lds_, st, ARGS);           \
    } while (0)

template <int AMODE>
static void launch_fwd(const FwdArgs &g, hipStream_t st)
{
    const int R = g.a.rows, Ci = g.w.ci, Co = g.w.co;
    if (AMODE == ACT_NONE && Ci == 3 && R > 64) {  // xyz input layer: streaming kernel, same stats layout (64 rows / block)
        static_assert(TileBig::BM == 64, "conv_in3_fwd_kernel writes one stats partial per 64 rows");
        hipLaunchKernelGGL(conv_in3_fwd_kernel, dim3((R + 63) / 64, (Co + 63) / 64), dim3(256), 0, st, R, Co, g.a.z, g.w.w,
                           g.bias, g.z, g.stats);
        return;
    }
    if (R <= 32) {
        if (Ci % 64 == 0 && Ci <= 512) {
            const size_t lds = ((size_t)64 * (Ci + 4) + 3 * 16 * 64 + 32 * 36) * sizeof(float);
            static bool attr_done = false;
            if (!attr_done) {
                (void)hipFuncSetAttribute((const void *)small_fwd_lds_kernel<AMODE>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                          (int)(((size_t)64 * 516 + 3 * 16 * 64 + 32 * 36) * sizeof(float)));
                attr_done = true;
            }
            hipLaunchKernelGGL((small_fwd_lds_kernel<AMODE>), dim3((Co + 31) / 32), dim3(256), lds, st, g);
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
        if (blk + 1 < nblk) fetch_b(blk_col(blk + 1));
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
                __builtin_amdgcn_sched_barrier(0);
                SN_WIDE_TERM(0, 2);
                SN_WIDE_TERM(2, 0);
                SN_WIDE_TERM(1, 1);
                SN_WIDE_TERM(0, 1);
                SN_WIDE_TERM(1, 0);
                SN_WIDE_TERM(0, 0);
                __builtin_amdgcn_sched_barrier(0);
#undef SN_WIDE_TERM
            }
            // the next block's weights (requested at the top of the iteration) go to the other buffer under the last MFMA groups; that
            // buffer's readers passed the barrier that ended iteration blk - 1
            if (kk == KS - 2 && blk + 1 < nblk) stage_b(Bs + ((blk + 1) & 1) * BUF);
            if (ep) {
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
        static bool attr = false;                                                                                                   \
        if (!attr) {                                                                                                                \
            if (hipFuncSetAttribute((const void *)linear_fwd_wide_pool_kernel<KK, SZ, AR>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                    (int)(2 * 3 * kWideBN * (KK + 8) * 2)) != hipSuccess)                                           \
                return sn_set_error(SN_ERR_UNSUPPORTED, "sn_linear_forward_maxpool_wide: cannot reserve LDS");                      \
            attr = true;                                                                                                            \
        }                                                                                                                           \
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
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute((const void *)pointnet_narrow_fwd_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
            hipFuncSetAttribute((const void *)pointnet_narrow_fwd_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return sn_set_error(SN_ERR_UNSUPPORTED, "sn_pointnet_narrow_forward: cannot reserve LDS");
        attr = true;
    }
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
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute((const void *)pointnet_narrow_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return sn_set_error(SN_ERR_UNSUPPORTED, "sn_pointnet_narrow_backward: cannot reserve LDS");
        attr = true;
    }
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
    static size_t attr_lds = 0;
    if (lds > attr_lds) {
        if (hipFuncSetAttribute((const void *)fc_chain_fwd_kernel<128, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
            hipFuncSetAttribute((const void *)fc_chain_fwd_kernel<256, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
            hipFuncSetAttribute((const void *)fc_chain_fwd_kernel<0, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return sn_set_error(SN_ERR_UNSUPPORTED, "sn_fc_chain_forward: %zu bytes of LDS refused", lds);
        attr_lds = lds;
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
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute((const void *)fc_chain_fwd_kernel<128, 3, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return sn_set_error(SN_ERR_UNSUPPORTED, "sn_fc_chain_forward_pool: %zu bytes of LDS refused", lds);
        attr_done = true;
    }
    g.rinv_rows = 1.0 / (double)B, g.unbias = B > 1 ? (double)B / (double)(B - 1) : 1.0;
    hipLaunchKernelGGL((fc_chain_fwd_kernel<128, 3, true>), dim3(8 * (H / 32)), dim3(256), lds, (hipStream_t)stream, g);
    SN_LAUNCH_CHECK();
    return 0;
}

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
    const size_t lds = ((size_t)3 * 32 * 260 + kRsFloats + 32 * 36) * sizeof(float);
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute((const void *)fc_chain_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return sn_set_error(SN_ERR_UNSUPPORTED, "sn_fc_chain_backward: %zu bytes of LDS refused", lds);
        attr_done = true;
    }
    // 128 blocks: b % 8 == 0 -> XCD 0; b / 8 = 0..7 the chain, 8..15 the weight-gradient workgroups
    hipLaunchKernelGGL(fc_chain_bwd_kernel, dim3(128), dim3(256), lds, (hipStream_t)stream, g);
    SN_LAUNCH_CHECK();
    return 0;
}

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
    for (int l = 1; l <= nlayers; ++l)
        if (channels[l] % 64 || channels[l] > 128) return 0;
    return 1;
}

// (behind the accumulators: room for the split weights of the nlayers - 1 GEMM layers, 128 x 128 x 3 bf16 each -- scratch of the
//  call, contents irrelevant between calls)
constexpr long long kWPlaneLL = (long long)128 * 128 * 3 * 2 / 8;
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
extern "C" long long sn_conv_stack_acc_sum_elems(int nlayers) { return nlayers > 0 ? (long long)nlayers * kFxLayer : 0; }
extern "C" long long sn_conv_stack_acc_elems(int nlayers)
{
    return nlayers > 0 ? (long long)nlayers * kFxLayer + (long long)(nlayers - 1) * kWPlaneLL : 0;
}

static int device_cus();
// tiles per workgroup from which the conv stack's forward GEMMs run as persistent kernels (0: never); a test / A-B hook
static int g_persist_min_tiles = 4;
extern "C" int sn_conv_stack_set_persist_min_tiles(int tiles)
{
    const int old = g_persist_min_tiles;
    g_persist_min_tiles = tiles;
    return old;
}
template <class TT, int KT, bool IN3A>
static int launch_fwd_persist(const FwdArgs &g, int ntiles, int tpw, int nwg, hipStream_t st)
{
    const size_t lds = sizeof(float) * ((size_t)(KT / BKX) * 3 * TT::BM * LDX + 2 * KT + (size_t)2 * TT::WR * 2 * TT::BN + (size_t)TT::WR * TT::WC * 16 * 36);
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute((const void *)linear_fwd_persist_kernel<TT, KT, IN3A, 2>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds) != hipSuccess)
            return sn_set_error(SN_ERR_UNSUPPORTED, "linear_fwd_persist_kernel: %zu bytes of LDS refused", lds);
        attr_done = true;
    }
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
        return sn_set_error(SN_ERR_UNSUPPORTED, "sn_conv_stack_forward_bn: needs N % 64 == 0 and 64 / 128 channels");
    SN_REQUIRE(x && W && gamma && beta && eps && momentum && z && coef && acc && pool_val && pool_idx, "null pointer");
    SN_REQUIRE((pooled && argsel && zsel) || (!pooled && !argsel && !zsel), "pooled / argsel / zsel: all or none");
    hipStream_t st = (hipStream_t)stream;
    const int R = B * N;
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
        __bf16 *base = reinterpret_cast<__bf16 *>(acc + (size_t)nlayers * kFxLayer);
        int nb = 0;
        for (int l = 1; l < nlayers; ++l) {
            const int li = l - 1, el = channels[l] * channels[l + 1];
            planes[l] = base + (size_t)li * kWPlaneLL * 4;
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
    hipLaunchKernelGGL(conv_in3_fwd_kernel, dim3(R / 64, channels[1] / 64), dim3(256), 0, st, R, channels[1], x, W[0],
                       bias ? bias[0] : nullptr, z[0], (float *)nullptr, acc, acc + (size_t)(nlayers - 1) * kFxLayer + kFxPoison, job);
    using T = TileBig;
    for (int l = 1; l < nlayers; ++l) {
        const int Ci = channels[l], Co = channels[l + 1];
        FwdArgs g{};
        g.a = make_act(z[l - 1], nullptr, R, Ci);
        g.w.w = W[l], g.w.co = Co, g.w.ci = Ci;
        g.bias = bias ? bias[l] : nullptr, g.z = z[l], g.stats = nullptr;
        g.acc_in = acc + (size_t)(l - 1) * kFxLayer, g.bn_prev = bn_of(l - 1), g.acc_out = acc + (size_t)l * kFxLayer;
        if (l >= 2) g.zero_ptr = acc + (size_t)(l - 2) * kFxLayer, g.zero_n = kFxLayer;
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
        if (pl && g.z && (l < nlayers - 1 || keys_pool) && (l > 1 || z1free) && (Ci == 64 || Co == 128) && R % 64 == 0) {
            const int ntiles = R / 64, per_cu = Co == 128 ? 1 : 2, nmax = device_cus() * per_cu;
            if (g_persist_min_tiles > 0 && ntiles >= g_persist_min_tiles * nmax) {
                const int tpw = (ntiles + nmax - 1) / nmax, nwg = (ntiles + tpw - 1) / tpw;
                if (l == 1) g.x3 = x, g.w3 = W[0], g.b3 = bias ? bias[0] : nullptr;
                int rc = 0;
                if (l == 1) rc = launch_fwd_persist<T, 64, true>(g, ntiles, tpw, nwg, st);
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
        if (Co == 128) {
            // 128 output channels: one 512-thread workgroup per 64 rows computes all of them -- the input tile is fetched
            // once instead of once per 64-column block, and half as many workgroups run the statistics prologue
            using TW = SN_FWD_TW;
            const dim3 grid(R / TW::BM, 1);
            const size_t lds = shaped_lds(lds_bytes<TW>() + (size_t)2 * Ci * sizeof(float), grid);
#define SN_FWD_FX(TT, KT_)                                                                                                \
    do {                                                                                                                  \
        if (pl) hipLaunchKernelGGL((linear_fwd_kernel<TT, true, ACT_BN_RELU_FX, KT_, SN_BF16X3 != 0>), grid, dim3(TT::THREADS), lds, st, g); \
        else hipLaunchKernelGGL((linear_fwd_kernel<TT, true, ACT_BN_RELU_FX, KT_>), grid, dim3(TT::THREADS), lds, st, g);  \
    } while (0)
            if (Ci == 128 && pl) SN_FWD_FX(TW, 128);
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
    long long *zp = nlayers >= 2 ? acc + (size_t)(nlayers - 2) * kFxLayer : nullptr;
    if (!pooled) {  // the last BatchNorm + the pool pick run as the first stage of sn_fc_chain_forward_pool
        SN_LAUNCH_CHECK();
        return 0;
    }
    if (keys_pool) {
        const int cpb = 8;  // clouds per workgroup
        hipLaunchKernelGGL(bn_finalize_pool_keys_kernel, dim3((B + cpb - 1) / cpb), dim3(256), 0, st, bn_of(nlayers - 1), B, Cn, cpb,
                           reinterpret_cast<const unsigned long long *>(pool_val), pooled, argsel, zsel,
                           acc + (size_t)(nlayers - 1) * kFxLayer, zp, zp ? kFxLayer : 0);
        SN_LAUNCH_CHECK();
        return 0;
    }
    hipLaunchKernelGGL(bn_finalize_pool_kernel, dim3((Cn + kChan - 1) / kChan), dim3(1024), 0, st, 0, Cn, (const float *)nullptr,
                       bn_of(nlayers - 1), B, N / 64, pool_val, pool_idx, pooled, argsel, zsel, acc + (size_t)(nlayers - 1) * kFxLayer, zp,
                       zp ? kFxLayer : 0);
    SN_LAUNCH_CHECK();
    return 0;
}

extern "C" int sn_linear_stats_blocks(int R) { return R > 64 ? (R + TileBig::BM - 1) / TileBig::BM : (R + TileSmall::BM - 1) / TileSmall::BM; }

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
static int device_cus()
{
    static int n = 0;
    if (n == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess &&
            hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0)
            n = v;
        else
            n = 256;  // MI355X
        (void)hipGetLastError();
    }
    return n;
}

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
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void *)conv_bwd_bx3_kernel<CI, CO, ZMODE, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void *)conv_bwd_bx3_kernel<CI, CO, ZMODE, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done = true;
    }
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
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void *)conv_bwd_bx3_kernel<128, 128, ZMODE, true, false, false, GZ, GP, GW, DM>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void *)conv_bwd_bx3_kernel<128, 128, ZMODE, false, false, false, GZ, GP, GW, DM>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done = true;
    }
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
    static bool attr_done = false;
    if (!attr_done) {  // more than 64 KB of dynamic LDS must be requested explicitly
        (void)hipFuncSetAttribute((const void *)conv_bwd_fused_kernel<CI, CO, ZMODE, true>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void *)conv_bwd_fused_kernel<CI, CO, ZMODE, false>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done = true;
    }
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
    const int nsplit = sn_linear_wgrad_splits(R, Ci, Co, db ? 1 : 0);
    int rps = (R + nsplit - 1) / nsplit;
    rps = ((rps + BK - 1) / BK) * BK;
    const bool fast = !db && coef_prev && dz_mode != DZ_PLAIN && R % TileBig::BM == 0 && Ci % TileBig::BN == 0 &&
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
    else
        hipLaunchKernelGGL((linear_bwd_kernel<TileBig, DZ_POOL, ACT_BN_RELU>), grid, dim3(TileBig::THREADS), lds, st, d, w, n_w, wgx, wgy, dgx);
    const int nred = (Co * Ci) % 4 == 0 ? (Co * Ci + kRedElems - 1) / kRedElems : (Co * Ci + 63) / 64;
    // (the statistics partials THIS route fills: one per TileBig row block -- at R == 64 sn_linear_stats_blocks counts TileSmall
    //  blocks, and summing that many read an unwritten block: wrong dgamma / dbeta / dbias of the layer below at exactly 64 rows)
    hipLaunchKernelGGL(post_bwd_kernel, dim3(nred + (Ci + kChan - 1) / kChan), dim3(1024), 0, st, nred, nsplit, Co, Ci, part, dW, dgx, Ci,
                       stats, bb);
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
        static bool attr_rz = false;
        if (!attr_rz) {
            (void)hipFuncSetAttribute((const void *)conv_bwd_bx3_kernel<64, 64, DZ_BN, true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            (void)hipFuncSetAttribute((const void *)conv_bwd_bx3_kernel<64, 64, DZ_BN, false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            attr_rz = true;
        }
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
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void *)SN_CBF_IN3<64, 64, DZ_BN, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void *)SN_CBF_IN3<64, 64, DZ_BN, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done = true;
    }
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
    for (int l = 1; l < nlayers; ++l)
        if (!conv_bwd_fused_shape(R, ch[l], ch[l + 1])) return false;
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

extern "C" int sn_bn_eval_coef(int C, const float *gamma, const float *beta, float eps, const float *running_mean,
                               const float *running_var, float *coef, sn_stream_t stream)
{
    SN_REQUIRE(C >= 1 && gamma && beta && running_mean && running_var && coef, "bad argument");
    hipLaunchKernelGGL(bn_eval_coef_kernel, dim3((C + 63) / 64), dim3(64), 0, (hipStream_t)stream, C, gamma, beta, eps,
                       running_mean, running_var, coef);
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

extern "C" int sn_pool_forward(int B, int N, int C, const float *z, const float *coef, float *pooled, int *argsel,
                               float *zsel, sn_stream_t stream)
{
    SN_REQUIRE(B >= 1 && N >= 1 && C >= 1 && z && coef && pooled && argsel && zsel, "bad argument");
    hipLaunchKernelGGL(pool_fwd_kernel, dim3(B, (C + 63) / 64), dim3(1024), 0, (hipStream_t)stream, N, C, z, coef, pooled,
                       argsel, zsel);
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

#ifdef SN_TIMELINE
extern "C" int sn_debug_fc_timeline(unsigned long long *host)
{
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(sn::sn_fc_tl_buf), sizeof(unsigned long long) * 2 * 16 * 32) != hipSuccess;
}

extern "C" int sn_debug_timeline(unsigned long long *host, int nblocks, int clear)
{
    if (clear == 2) return hipMemcpyFromSymbol(host, HIP_SYMBOL(sn::sn_hw_buf), sizeof(unsigned) * 8 * (size_t)nblocks) != hipSuccess;
    if (clear) {
        void *p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(sn::sn_tl_buf)) != hipSuccess) return 1;
        return hipMemset(p, 0, sizeof(unsigned long long) * 16384 * 16) != hipSuccess;
    }
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(sn::sn_tl_buf), sizeof(unsigned long long) * 16 * (size_t)nblocks) != hipSuccess;
}
#endif
