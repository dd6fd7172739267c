// mlp_device.h -- device-side building blocks shared by the PointNet MLP translation units (pointnet_mlp.hip: forward layer GEMMs,
// conv stack, BatchNorm, pooling; pointnet_mlp_backward.hip: their backward; fc_chain.hip: the FC head as one resident-workgroup
// launch per direction; task_network.hip: the task network's BatchNorm-free extractor and skinny trunk): operand loaders, the fp32 and split-bf16 GEMM tile cores,
// fixed-point BatchNorm statistics, BatchNorm finalisation / backward expressions, pool keys, LDS-only barrier and the
// cross-wave reduce-scatter.  Everything here is inline device code or a template: each translation unit instantiates what it
// launches.  Debug builds (-DSN_TIMELINE: tools/build_timeline_lib.sh compiles the four units as ONE so that the stamp buffers
// exist once) also carry the phase-stamp buffers and their read-out entries.
#pragma once
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
constexpr int kFxRow = 128;                           // channels per row (a layer of 129 .. 256 channels: TWO blocks back to back,
                                                      //  channel c in block c >> 7 -- sn_conv_stack_forward_bn's wide layout)
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

// Staging map of the K chunks: item f (4 consecutive k of one row: 8 bytes per plane) -> row.  The 16 lanes of a ds_write_b64 lane
// group hold two rows; rows x and x + 1 are 80 bytes apart (banks 20-35 wrap onto 0-3: a 2-way conflict on every store of the three
// planes -- the 20-29 % conflict share of the forward kernels' LDS cycles in profiles/r06/lds_map.txt).  SN_BX3_ROWMAP=1 pairs rows x
// and x + 4 instead (320 bytes apart: disjoint halves of the stores' 32 banks; a bijection inside every block of 8 rows) -- built and
// measured in round 6: bit-identical, and NOT faster (B = 2048: 548.7 / 324.6 / 215.2 / 185.4 us against 551.0 / 330.0 / 214.1 / 183.1;
// B = 32 equal): LDS stores are paced by the register transfer, not by the array (MI355X_MICROARCH.md, LDS).  Left off.
#ifndef SN_BX3_ROWMAP
#define SN_BX3_ROWMAP 0
#endif
__device__ __forceinline__ constexpr int bx3_row(int f)
{
    return SN_BX3_ROWMAP ? (((f >> 6) << 3) + ((f >> 4) & 3) + (((f >> 3) & 1) << 2)) : f / (BKX / 4);
}
static_assert(BKX == 32, "bx3_row: 8 items per row");
// the same for 16-byte items (pre-split weight planes: 4 items per row, ds_write_b128 in lane groups of 8 = two rows)
__device__ __forceinline__ constexpr int bx3_row8(int f)
{
    return SN_BX3_ROWMAP ? (((f >> 5) << 3) + ((f >> 3) & 3) + (((f >> 2) & 1) << 2)) : f / (BKX / 8);
}

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
        ra[q] = fa(bx3_row(f), k0 + (f % (BKX / 4)) * 4);
    }
#pragma unroll
    for (int q = 0; q < Bx3<T>::B4; ++q) {
        const int f = tid + q * T::THREADS;
        rb[q] = fb(bx3_row(f), k0 + (f % (BKX / 4)) * 4);
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
            stage_split<T::BN>(Bp, bx3_row(f), (f % (BKX / 4)) * 4, rb[q]);
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
        stage_split<T::BM>(Ap, bx3_row(f), k4, xa(ra[q], k0 + k4));
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
        const int f = tid + q * T::THREADS, x = bx3_row8(f), k8 = (f % (BKX / 8)) * 8;
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
        const int f = tid + q * T::THREADS, x = bx3_row8(f), k8 = (f % (BKX / 8)) * 8;
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
        ra[q] = fa(bx3_row(f), k0 + (f % (BKX / 4)) * 4);
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
    // small_fwd_lds_kernel with bn.coef: also the NORMALISED output y = z scale + shift, no activation, (R, Co) -- the head's
    // output layer of the classification sampler (classification/models/samplenet_model.py:100-108: fc14b, bn, activation_fn=None)
    float *bn_y;
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

constexpr int KP = 32;  // R <= 32 kernels: k values per lane per pass

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains the wave's GLOBAL loads and stores
// (s_waitcnt vmcnt(0) in front of s_barrier): every barrier of the chain kernels then exposed the full latency of the operand
// prefetches in flight across it (timestamps: ~1.7 us per "MFMA phase" that holds 0.4 us of MFMAs).  Global data never
// crosses these barriers (hand-offs are drained explicitly before their arrival atomics).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

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

}  // namespace sn

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
