// emd.hip -- approx_match / match_cost / match_cost_grad for gfx950.
//
// Reference: classification/structural_losses/tf_approxmatch_g.cu:1-295 (one 512-thread block per
// cloud, <<<32,512>>>; `match` is read-modify-written in global memory at each of the 10 levels).
//
// Re-design for MI355X.  The auction is kept arithmetically identical -- every per-point sum runs
// sequentially over the other cloud in ascending index order, in fp32, exactly as each reference
// thread accumulates it -- but it is spread over the whole chip and `match` is written ONCE:
//   * one THREAD per point, many workgroups per cloud; the three block-synchronous passes of a
//     level become two kernels (the level's pass 1 is fused with the previous level's pass 3,
//     both being sums over l for a fixed k), i.e. 21 launches instead of a 30-phase single block;
//   * the other cloud's coordinates + per-point weight are staged through LDS as float4 tiles and
//     read back as wave-uniform broadcasts (ds_read_b128, conflict-free);
//   * the per-level ratio vectors (10 x (n+m) floats per cloud) are kept in the workspace, and
//     match[l,k] = sum_levels exp(level d) ratioL_lev[k] ratioR_lev[l] is materialised by one
//     streaming kernel -- the same terms added in the same order as the reference's `+=`, so the
//     result is identical, at 1/20 of the HBM traffic (write-once instead of 10 x read+write).
#include <algorithm>

#include "sn_common.h"

#pragma clang fp contract(off)

namespace sn {

constexpr int kLevels = 10;  // j = 7 .. -2  (tf_approxmatch_g.cu:21)
constexpr int kTile = 1024;  // points of the other cloud staged per LDS tile (16 KiB)

__host__ __device__ inline float emd_level(int li)
{
    // li = 0..9  <->  j = 7..-2 ; level = -4^j, 0 at j == -2   (tf_approxmatch_g.cu:22-25)
    const int j = 7 - li;
    if (j == -2) return 0.0f;
    float p = 1.0f;
    if (j >= 0)
        for (int t = 0; t < j; ++t) p *= 4.0f;
    else
        p = 0.25f;  // j == -1
    return -p;
}

__device__ __forceinline__ float emd_sq(float ax, float ay, float az, float bx, float by, float bz)
{
    // (x2-x1)*(x2-x1)+(y2-y1)*(y2-y1)+(z2-z1)*(z2-z1)   (tf_approxmatch_g.cu:48)
    const float dx = bx - ax, dy = by - ay, dz = bz - az;
    return (dx * dx + dy * dy) + dz * dz;
}

// exp for the auction.  The reference op uses __expf (tf_approxmatch_g.cu:49,52,97,151) -- v_exp_f32(x log2 e) on this hardware;
// its CPU twin (and the oracle) use expf.  Two forms, a template parameter of every sweep:
//   FAST = false  (sn_approxmatch, sn_emd_loss: whatever hands out or must reproduce the MATCH matrix -- emd_matching takes its
//           argmax): the auction amplifies exp rounding through its 10 levels (plain v_exp_f32 moves ~2 % of the argmax picks
//           and breaks the 5e-4 per-entry bar held on `match`), so the hardware exponential gets a compensated argument:
//           y = x log2(e) split into its rounded value and the rounding residual (one fma + the low word of log2 e), 2^y from
//           v_exp_f32 (1 ulp), the residual applied as 1 + r ln 2 -- ~2 ulp overall at 6 packed operations + 2 v_exp per pair;
//   FAST = true   (sn_emd_loss_fast: the LOSS -- cost and its gradients, bar 1e-5 on the cost): the reference's own
//           __expf, 1 packed multiply + 2 v_exp per pair.
// Every level is -4^j, a power of two, so x = level d is exact and fl(x c) == fl(d (level c)) for any constant c: the level is
// folded into the constants (level log2e_hi, level log2e_lo) once per kernel -- same bits, one multiply less per exponential.
//
// The sweeps are VALU-issue-bound, so they work on PAIRS of points with gfx950's packed fp32 instructions (v_pk_mul / v_pk_add /
// v_pk_fma_f32: two values in one 64-bit register pair per issue).  They are written as inline assembly with EARLY-CLOBBER
// destinations, i.e. the destination pair never overlaps a source pair: the library is otherwise built without packed fp32
// arithmetic (build.py, DESIGN.md 6c: with a second process on the GPU, compiler-packed code whose destination pair aliased a
// source returned wrong low halves -- a wave restored in the middle of such a two-pass instruction re-executes it on a half it
// has already overwritten); an instruction whose sources survive it computes the same result however often it is replayed.
// This translation unit is compiled with the packed feature on and the SLP vectoriser off, and
// tests/test_cabi_and_host.py::test_device_code_carries_no_packed_fp32_arithmetic checks in the disassembly that every packed
// instruction left in emd.o has a destination disjoint from its sources (and that no other object has any).
typedef float f2v __attribute__((ext_vector_type(2)));
#if defined(SN_EMD_SCALAR_F32) && SN_EMD_SCALAR_F32
// Build switch (build.py: SAMPLENET_AMD_EMD_SCALAR=1; this unit is then compiled like the others, without the packed feature):
// the same pair helpers element by element -- each element rounds exactly as the packed lane does, results are bit-identical,
// the sweeps issue two scalar VALU instructions where the default build issues one packed one.
__device__ __forceinline__ f2v pk_mul(f2v a, f2v b) { return (f2v){a.x * b.x, a.y * b.y}; }
__device__ __forceinline__ f2v pk_add(f2v a, f2v b) { return (f2v){a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ f2v pk_sub(f2v a, f2v b) { return (f2v){a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ f2v pk_fma(f2v a, f2v b, f2v c) { return (f2v){__builtin_fmaf(a.x, b.x, c.x), __builtin_fmaf(a.y, b.y, c.y)}; }
__device__ __forceinline__ f2v pk_fms(f2v a, f2v b, f2v c) { return (f2v){__builtin_fmaf(a.x, b.x, -c.x), __builtin_fmaf(a.y, b.y, -c.y)}; }
#else
__device__ __forceinline__ f2v pk_mul(f2v a, f2v b)
{
    f2v d;
    asm("v_pk_mul_f32 %0, %1, %2" : "=&v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ f2v pk_add(f2v a, f2v b)
{
    f2v d;
    asm("v_pk_add_f32 %0, %1, %2" : "=&v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ f2v pk_sub(f2v a, f2v b)  // a - b
{
    f2v d;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=&v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ f2v pk_fma(f2v a, f2v b, f2v c)  // a b + c, one rounding
{
    f2v d;
    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=&v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ f2v pk_fms(f2v a, f2v b, f2v c)  // a b - c, one rounding
{
    f2v d;
    asm("v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=&v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
#endif
__device__ __forceinline__ f2v splat(float v) { return (f2v){v, v}; }

constexpr float kLog2eHi = 1.44269502162933349609375f, kLog2eLo = 1.925963033500966e-8f, kLn2 = 0.693147180559945309f;
// the per-level constants of the exponential: (level log2e_hi, level log2e_lo) as pairs (exact scalings: the level is a power of two)
struct EmdLev {
    f2v hi, lo;
};
__device__ __forceinline__ EmdLev emd_lev(float level) { return EmdLev{splat(level * kLog2eHi), splat(level * kLog2eLo)}; }
// exp(level d2) for a pair of squared distances
template <bool FAST>
__device__ __forceinline__ f2v emd_exp2(f2v d2, const EmdLev &L)
{
    const f2v y = pk_mul(d2, L.hi);
    f2v e = {__builtin_amdgcn_exp2f(y.x), __builtin_amdgcn_exp2f(y.y)};
    // gfx950 needs one wait state between a transcendental and a VALU instruction that reads its result; the compiler pads that
    // hazard for its own instructions only -- it cannot see that the inline-assembly multiply behind this is one (without the
    // pad the multiply read the register's OLD value: negative "exponentials", NaN losses)
    asm("s_nop 0" : "+v"(e));
    if (FAST) return e;
    const f2v r = pk_add(pk_fms(d2, L.hi, y), pk_mul(d2, L.lo));
    return pk_mul(e, pk_fma(r, splat(kLn2), splat(1.0f)));
}
// squared distances of one point (given as splat pairs) to a PAIR of points: (b - a)^2 summed as emd_sq does, per lane
__device__ __forceinline__ f2v emd_sq2(f2v ax, f2v ay, f2v az, f2v bx, f2v by, f2v bz)
{
    const f2v dx = pk_sub(bx, ax), dy = pk_sub(by, ay), dz = pk_sub(bz, az);
    return pk_add(pk_add(pk_mul(dx, dx), pk_mul(dy, dy)), pk_mul(dz, dz));
}

// Workspace layout per cloud (floats): remainL[n] remainR[m] ratioL[kLevels][n] ratioR[kLevels][m]
__host__ __device__ inline size_t emd_ws_floats(int n, int m) { return (size_t)(n + m) * (1 + kLevels); }

// Kernel A(li): thread per xyz1 point k.
//   if li > 0 : pass 3 of level li-1 -> remainL[k] = max(0, remainL[k] - sum_l exp(lev' d) ratioL'[k] ratioR'[l])
//   then      : pass 1 of level li   -> ratioL[li][k] = remainL[k] / (1e-9 + sum_l exp(lev d) remainR[l])
// (pass 3 of the last level only updates remainL, which nothing reads afterwards: it is not run.)
// li == 0 also initialises remainL (and, by its first block column, remainR).
// Segments (round 6): at BASELINE configs[3] a level pass is 400 workgroups of equal, long work on 256 CUs -- 1.56 waves per SIMD, i.e.
// some SIMDs run two waves while others run one and the launch lasts as long as the former (78 % of the issue slots used).  With
// nseg > 1 the other cloud is cut into nseg ranges (gridDim.z): a workgroup sweeps one range and leaves its partial sums in seg[];
// the LAST workgroup of a (cloud, point block) to arrive -- a relaxed counter behind a device-scope fence -- adds the nseg partials in
// ascending order and finishes the pass: thousands of short workgroups that the dispatcher balances by itself, sums still in a fixed
// order (deterministic; the order differs from the one-range sweep's only in where the partial sums are cut).  nseg == 1: the
// one-range code path, unchanged.
struct EmdSeg {
    float *psum;        // pass k: [b][nseg][2][n] (sum3 | sum1 partials); pass l: [b][nseg][m]
    unsigned *counter;  // [b][gridDim.x], zero between launches
    int nseg, len;      // ranges of `len` points of the other cloud
};
// (partials cross workgroups as write-through stores / sc1 loads around a relaxed counter, as the skinny layers' slice partials do
//  (task_network.hip): no fences -- an agent-scope fence writes back and invalidates the whole L2 of the XCD, and thousands of
//  workgroups doing that per launch doubled the auction's time)
__device__ __forceinline__ void emd_store_sc1(float *p, float v) { asm volatile("global_store_dword %0, %1, off sc1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ float emd_load_sc1(const float *p)
{
    float v;
    asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ bool emd_seg_last(const EmdSeg &sg, int b)
{
    __shared__ int s_last;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this workgroup's partial sums have left before its arrival
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned *c = sg.counter + (size_t)b * gridDim.x + blockIdx.x;
        const unsigned t = __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = t == (unsigned)(sg.nseg - 1);
        if (s_last) __hip_atomic_store(c, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // armed for the next launch
    }
    __syncthreads();
    return s_last != 0;
}

template <bool FAST>
__global__ void __launch_bounds__(256) emd_pass_k_kernel(int n, int m, int li, const float *__restrict__ xyz1,
                                                         const float *__restrict__ xyz2, float *__restrict__ ws,
                                                         float multiL, float multiR, EmdSeg sg)
{
    __shared__ float4 tile[kTile];
    const int b = blockIdx.y;
    const int lbeg = sg.nseg > 1 ? (int)blockIdx.z * sg.len : 0, lstop = sg.nseg > 1 ? min(m, lbeg + sg.len) : m;
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    const float *X1 = xyz1 + (size_t)b * n * 3;
    const float *X2 = xyz2 + (size_t)b * m * 3;
    float *base = ws + (size_t)b * emd_ws_floats(n, m);
    float *remainL = base, *remainR = base + n;
    float *ratioL = base + n + m, *ratioR = ratioL + (size_t)kLevels * n;
    const bool do3 = li > 0;
    const float lev3 = do3 ? emd_level(li - 1) : 0.f, lev1 = emd_level(li);
    const float *rR3 = ratioR + (size_t)(do3 ? li - 1 : 0) * m;

    float x1 = 0, y1 = 0, z1 = 0, rl = 0;
    if (k < n) {
        x1 = X1[k * 3 + 0], y1 = X1[k * 3 + 1], z1 = X1[k * 3 + 2];
        if (do3) rl = ratioL[(size_t)(li - 1) * n + k];
    }
    // ONE sweep over the other cloud feeds both sums (the squared distance is shared; each sum still adds its terms in
    // ascending l, so the results are those of the two separate sweeps):
    //   sum3 = sum_l exp(lev3 d) ratioL'[k] ratioR'[l]   (pass 3 of the previous level)      -> remainL
    //   sum1 = 1e-9 + sum_l exp(lev1 d) remainR[l]       (pass 1 of this level; remainR was finished by the previous
    //                                                     level's pass 2, it does not depend on remainL)
    // (Two threads per point -- half sums combined at the end -- to raise the ~1.5 waves per SIMD at B = 50 were measured
    //  SLOWER, 4.2 vs 3.9 ms: the sweep is bound by VALU issue, not by latency.)
    // LDS tiles hold PAIRS of points in register-pair order: tileA[q] = {x_l, x_l+1, y_l, y_l+1}, tileB[q] = {z_l, z_l+1,
    // w_l, w_l+1} (w = remainR), tileC[q] = {w3_l, w3_l+1} (w3 = ratioR'); an odd tail is padded with a zero-weight point.
    float4 *tileA = tile, *tileB = tile + kTile / 2;
    __shared__ float2 tileC[kTile / 2];
    float sum3 = 0.f, sum1 = sg.nseg > 1 ? 0.f : 1e-9f;
    for (int l0 = lbeg; l0 < lstop; l0 += kTile) {
        const int lend = min(lstop, l0 + kTile) - l0, npair = (lend + 1) >> 1;
        __syncthreads();
        for (int q = threadIdx.x; q < npair; q += blockDim.x) {
            const int la = l0 + 2 * q, lb = la + 1;
            const bool okb = 2 * q + 1 < lend;
            const float xb = okb ? X2[lb * 3 + 0] : 0.f, yb = okb ? X2[lb * 3 + 1] : 0.f, zb = okb ? X2[lb * 3 + 2] : 0.f;
            tileA[q] = make_float4(X2[la * 3 + 0], xb, X2[la * 3 + 1], yb);
            tileB[q] = make_float4(X2[la * 3 + 2], zb, li == 0 ? multiR : remainR[la], okb ? (li == 0 ? multiR : remainR[lb]) : 0.f);
            if (do3) tileC[q] = make_float2(rR3[la], okb ? rR3[lb] : 0.f);
        }
        __syncthreads();
        const f2v px = splat(x1), py = splat(y1), pz = splat(z1), prl = splat(rl);
        const EmdLev L3 = emd_lev(lev3), L1 = emd_lev(lev1);
        // four pairs per trip, written out (the unroller leaves loops with inline assembly alone): the packed work of the four
        // is independent and interleaves; the sums still take their terms one by one in ascending l
        if (do3) {
            auto body = [&](int q) __attribute__((always_inline)) {
                const float4 ta = tileA[q], tb = tileB[q];
                const float2 tc = tileC[q];
                const f2v d2 = emd_sq2(px, py, pz, (f2v){ta.x, ta.y}, (f2v){ta.z, ta.w}, (f2v){tb.x, tb.y});
                const f2v w3 = pk_mul(pk_mul(emd_exp2<FAST>(d2, L3), prl), (f2v){tc.x, tc.y});
                const f2v w1 = pk_mul(emd_exp2<FAST>(d2, L1), (f2v){tb.z, tb.w});
                sum3 += w3.x;
                sum3 += w3.y;
                sum1 += w1.x;
                sum1 += w1.y;
            };
            int q = 0;
            for (; q + 4 <= npair; q += 4) body(q), body(q + 1), body(q + 2), body(q + 3);
            for (; q < npair; ++q) body(q);
        } else {
            auto body = [&](int q) __attribute__((always_inline)) {
                const float4 ta = tileA[q], tb = tileB[q];
                const f2v d2 = emd_sq2(px, py, pz, (f2v){ta.x, ta.y}, (f2v){ta.z, ta.w}, (f2v){tb.x, tb.y});
                const f2v w1 = pk_mul(emd_exp2<FAST>(d2, L1), (f2v){tb.z, tb.w});
                sum1 += w1.x;
                sum1 += w1.y;
            };
            int q = 0;
            for (; q + 4 <= npair; q += 4) body(q), body(q + 1), body(q + 2), body(q + 3);
            for (; q < npair; ++q) body(q);
        }
    }
    if (sg.nseg > 1) {
        float *ps = sg.psum + ((size_t)b * sg.nseg + blockIdx.z) * 2 * n;
        if (k < n) emd_store_sc1(ps + k, sum3), emd_store_sc1(ps + n + k, sum1);
        if (!emd_seg_last(sg, b)) return;
        sum3 = 0.f, sum1 = 1e-9f;
        if (k < n)
            for (int q = 0; q < sg.nseg; ++q) {
                const float *pq = sg.psum + ((size_t)b * sg.nseg + q) * 2 * n;
                sum3 += emd_load_sc1(pq + k), sum1 += emd_load_sc1(pq + n + k);
            }
    }
    float remL = multiL;
    if (k < n) {
        if (do3) remL = fmaxf(0.0f, remainL[k] - sum3);
        if (do3 || li == 0) remainL[k] = remL;
    }
    if (li == 0 && blockIdx.x == 0)
        for (int l = threadIdx.x; l < m; l += blockDim.x) remainR[l] = multiR;
    if (k < n) ratioL[(size_t)li * n + k] = remL / sum1;
}

// Kernel B(li): thread per xyz2 point l -- pass 2 of level li (tf_approxmatch_g.cu:75-110).
template <bool FAST>
__global__ void __launch_bounds__(256) emd_pass_l_kernel(int n, int m, int li, const float *__restrict__ xyz1,
                                                         const float *__restrict__ xyz2, float *__restrict__ ws, EmdSeg sg)
{
    __shared__ float4 tile[kTile];
    const int b = blockIdx.y;
    const int kbeg = sg.nseg > 1 ? (int)blockIdx.z * sg.len : 0, kstop = sg.nseg > 1 ? min(n, kbeg + sg.len) : n;
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    const float *X1 = xyz1 + (size_t)b * n * 3;
    const float *X2 = xyz2 + (size_t)b * m * 3;
    float *base = ws + (size_t)b * emd_ws_floats(n, m);
    float *remainR = base + n;
    const float *ratioL = base + n + m + (size_t)li * n;
    float *ratioR = base + n + m + (size_t)kLevels * n + (size_t)li * m;
    const float level = emd_level(li);
    float x2 = 0, y2 = 0, z2 = 0;
    if (l < m) x2 = X2[l * 3 + 0], y2 = X2[l * 3 + 1], z2 = X2[l * 3 + 2];
    float sumr = 0.f;
    float4 *tileA = tile, *tileB = tile + kTile / 2;  // pairs of xyz1 points, as in emd_pass_k_kernel (w = ratioL)
    const EmdLev Lv = emd_lev(level);
    const f2v px = splat(x2), py = splat(y2), pz = splat(z2);
    for (int k0 = kbeg; k0 < kstop; k0 += kTile) {
        const int kend = min(kstop, k0 + kTile) - k0, npair = (kend + 1) >> 1;
        __syncthreads();
        for (int q = threadIdx.x; q < npair; q += blockDim.x) {
            const int ka = k0 + 2 * q, kb = ka + 1;
            const bool okb = 2 * q + 1 < kend;
            const float xb = okb ? X1[kb * 3 + 0] : 0.f, yb = okb ? X1[kb * 3 + 1] : 0.f, zb = okb ? X1[kb * 3 + 2] : 0.f;
            tileA[q] = make_float4(X1[ka * 3 + 0], xb, X1[ka * 3 + 1], yb);
            tileB[q] = make_float4(X1[ka * 3 + 2], zb, ratioL[ka], okb ? ratioL[kb] : 0.f);
        }
        __syncthreads();
        auto body = [&](int q) __attribute__((always_inline)) {
            const float4 ta = tileA[q], tb = tileB[q];
            // (x2 - x1)^2 ...: the pair is the FIRST operand of emd_sq here; squares are sign-symmetric, so the packed form
            // (pair - point) gives the same values
            const f2v d2 = emd_sq2(px, py, pz, (f2v){ta.x, ta.y}, (f2v){ta.z, ta.w}, (f2v){tb.x, tb.y});
            const f2v w = pk_mul(emd_exp2<FAST>(d2, Lv), (f2v){tb.z, tb.w});
            sumr += w.x;
            sumr += w.y;
        };
        int q = 0;
        for (; q + 4 <= npair; q += 4) body(q), body(q + 1), body(q + 2), body(q + 3);
        for (; q < npair; ++q) body(q);
    }
    if (sg.nseg > 1) {
        float *ps = sg.psum + ((size_t)b * sg.nseg + blockIdx.z) * m;
        if (l < m) emd_store_sc1(ps + l, sumr);
        if (!emd_seg_last(sg, b)) return;
        sumr = 0.f;
        if (l < m)
            for (int q = 0; q < sg.nseg; ++q) sumr += emd_load_sc1(sg.psum + ((size_t)b * sg.nseg + q) * m + l);
    }
    if (l < m) {
        const float rr = remainR[l];
        sumr *= rr;
        const float consumption = fminf(rr / (sumr + 1e-9f), 1.0f);
        ratioR[l] = consumption * rr;
        remainR[l] = fmaxf(0.0f, rr - sumr);
    }
}

// match[l,k] = sum over levels (in level order) of exp(level d_kl) * ratioL[lev][k] * ratioR[lev][l]
__global__ void __launch_bounds__(256) emd_materialize_kernel(int n, int m, const float *__restrict__ xyz1,
                                                              const float *__restrict__ xyz2,
                                                              const float *__restrict__ ws, float *__restrict__ match)
{
    const int b = blockIdx.z;
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    const int l0 = blockIdx.y * 16;
    const float *X1 = xyz1 + (size_t)b * n * 3;
    const float *X2 = xyz2 + (size_t)b * m * 3;
    const float *base = ws + (size_t)b * emd_ws_floats(n, m);
    const float *ratioL = base + n + m, *ratioR = ratioL + (size_t)kLevels * n;
    if (k >= n) return;
    const float x1 = X1[k * 3 + 0], y1 = X1[k * 3 + 1], z1 = X1[k * 3 + 2];
    float rl[kLevels];
#pragma unroll
    for (int li = 0; li < kLevels; ++li) rl[li] = ratioL[(size_t)li * n + k];
    float *Mb = match + (size_t)b * n * m;
    for (int l = l0; l < min(m, l0 + 16); ++l) {  // l is block-uniform: scalar loads
        const float d2 = emd_sq(x1, y1, z1, X2[l * 3 + 0], X2[l * 3 + 1], X2[l * 3 + 2]);
        float acc = 0.f;
        const f2v dd = splat(d2);
#pragma unroll
        for (int li = 0; li < kLevels; li += 2) {  // two levels per packed issue; the terms still enter acc in level order
            const EmdLev L2{(f2v){emd_level(li) * kLog2eHi, emd_level(li + 1) * kLog2eHi}, (f2v){emd_level(li) * kLog2eLo, emd_level(li + 1) * kLog2eLo}};
            const f2v w = pk_mul(pk_mul(emd_exp2<false>(dd, L2), (f2v){rl[li], rl[li + 1]}),
                                 (f2v){ratioR[(size_t)li * m + l], ratioR[(size_t)(li + 1) * m + l]});
            acc += w.x;
            acc += w.y;
        }
        Mb[(size_t)l * n + k] = acc;
    }
}

// cost[b] = sum_{k,l} match[l,k] ||x1_k - x2_l||.  Thread per k accumulates over l (as each reference
// thread does, tf_approxmatch_g.cu:183-213), fixed-order block tree, then one partial per block;
// partials are combined in index order by the last kernel -> deterministic.
__global__ void __launch_bounds__(256) emd_cost_partial_kernel(int n, int m, const float *__restrict__ xyz1,
                                                               const float *__restrict__ xyz2,
                                                               const float *__restrict__ match,
                                                               float *__restrict__ partial)
{
    __shared__ float red[256];
    const int b = blockIdx.y;
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    const float *X1 = xyz1 + (size_t)b * n * 3;
    const float *X2 = xyz2 + (size_t)b * m * 3;
    const float *Mb = match + (size_t)b * n * m;
    float sub = 0.f;
    if (k < n) {
        const float x1 = X1[k * 3 + 0], y1 = X1[k * 3 + 1], z1 = X1[k * 3 + 2];
        for (int l = 0; l < m; ++l) {
            const float d = sqrtf(emd_sq(x1, y1, z1, X2[l * 3 + 0], X2[l * 3 + 1], X2[l * 3 + 2]));
            sub += d * Mb[(size_t)l * n + k];
        }
    }
    red[threadIdx.x] = sub;
    for (int s = 128; s > 0; s >>= 1) {
        __syncthreads();
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    }
    if (threadIdx.x == 0) partial[(size_t)b * gridDim.x + blockIdx.x] = red[0];
}

__global__ void emd_cost_final_kernel(int nb, int nparts, const float *__restrict__ partial, float *__restrict__ cost)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nb) return;
    float acc = 0.f;
    for (int p = 0; p < nparts; ++p) acc += partial[(size_t)b * nparts + p];
    cost[b] = acc;
}

// grad1[k] = sum_l match[l,k] (x1_k - x2_l) rsqrt(max(d2, 1e-20))     (tf_approxmatch_g.cu:270-291)
__global__ void __launch_bounds__(256) emd_grad1_kernel(int n, int m, const float *__restrict__ xyz1,
                                                        const float *__restrict__ xyz2,
                                                        const float *__restrict__ match, float *__restrict__ grad1)
{
    const int b = blockIdx.y;
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const float *X1 = xyz1 + (size_t)b * n * 3;
    const float *X2 = xyz2 + (size_t)b * m * 3;
    const float *Mb = match + (size_t)b * n * m;
    const float x1 = X1[k * 3 + 0], y1 = X1[k * 3 + 1], z1 = X1[k * 3 + 2];
    float dx = 0, dy = 0, dz = 0;
    for (int l = 0; l < m; ++l) {
        const float ex = x1 - X2[l * 3 + 0], ey = y1 - X2[l * 3 + 1], ez = z1 - X2[l * 3 + 2];
        const float d = Mb[(size_t)l * n + k] * rsqrtf(fmaxf((ex * ex + ey * ey) + ez * ez, 1e-20f));
        dx += ex * d;
        dy += ey * d;
        dz += ez * d;
    }
    grad1[((size_t)b * n + k) * 3 + 0] = dx;
    grad1[((size_t)b * n + k) * 3 + 1] = dy;
    grad1[((size_t)b * n + k) * 3 + 2] = dz;
}

// grad2[l] = sum_k match[l,k] (x2_l - x1_k) rsqrt(max(d2, 1e-20)): a wave per l, lanes stride over k
// (coalesced match row), fixed-order butterfly                         (tf_approxmatch_g.cu:229-269)
__global__ void __launch_bounds__(256) emd_grad2_kernel(int n, int m, const float *__restrict__ xyz1,
                                                        const float *__restrict__ xyz2,
                                                        const float *__restrict__ match, float *__restrict__ grad2)
{
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int l = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (l >= m) return;
    const float *X1 = xyz1 + (size_t)b * n * 3;
    const float *X2 = xyz2 + (size_t)b * m * 3;
    const float *Mrow = match + (size_t)b * n * m + (size_t)l * n;
    const float x2 = X2[l * 3 + 0], y2 = X2[l * 3 + 1], z2 = X2[l * 3 + 2];
    float sx = 0, sy = 0, sz = 0;
    for (int k = lane; k < n; k += 64) {
        const float ex = x2 - X1[k * 3 + 0], ey = y2 - X1[k * 3 + 1], ez = z2 - X1[k * 3 + 2];
        const float d = Mrow[k] * rsqrtf(fmaxf((ex * ex + ey * ey) + ez * ez, 1e-20f));
        sx += ex * d;
        sy += ey * d;
        sz += ez * d;
    }
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) {
        sx += __shfl_xor(sx, s);
        sy += __shfl_xor(sy, s);
        sz += __shfl_xor(sz, s);
    }
    if (lane == 0) {
        grad2[((size_t)b * m + l) * 3 + 0] = sx;
        grad2[((size_t)b * m + l) * 3 + 1] = sy;
        grad2[((size_t)b * m + l) * 3 + 2] = sz;
    }
}

// ------------------------------------------------------------------------------------------------
// EMD loss without the match matrix (what reconstruction/src/samplenet_pointnet_ae.py:129-131 wants: match_cost(approx_match)
// and its gradient): match[l,k] = sum_levels exp(level d) ratioL_lev[k] ratioR_lev[l] is re-evaluated from the per-level ratio
// vectors (packed: two levels per issue) inside the cost / gradient sweeps instead of being written (839 MB at B = 50,
// 2048 x 2048) and read back three times.  The exponentials are evaluated twice (once per reduction axis); the sweeps
// are VALU-bound like the level passes.
//   emd_loss_k_kernel: thread per xyz1 point k, sequential over l (the order of emd_cost_partial_kernel / emd_grad1_kernel:
//                      cost and grad1 are bit-identical to sn_matchcost / sn_matchcost_grad on the materialised match)
//   emd_loss_l_kernel: thread per xyz2 point l, sequential over k -> grad2
// LDS tile of the other cloud: per point {x, y, z, pad} + its 10 ratios.
// ------------------------------------------------------------------------------------------------
constexpr int kLossTile = 256;
template <bool KSIDE, bool FAST>
__global__ void __launch_bounds__(256) emd_loss_sweep_kernel(int n, int m, const float *__restrict__ xyz1,
                                                             const float *__restrict__ xyz2, const float *__restrict__ ws,
                                                             float *__restrict__ partial, float *__restrict__ grad)
{
    __shared__ float4 tpt[kLossTile];
    __shared__ float trat[kLevels][kLossTile];
    __shared__ float red[256];
    const int b = blockIdx.y;
    const int nself = KSIDE ? n : m, nother = KSIDE ? m : n;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;  // k (KSIDE) or l
    const float *Xs = (KSIDE ? xyz1 : xyz2) + (size_t)b * nself * 3;
    const float *Xo = (KSIDE ? xyz2 : xyz1) + (size_t)b * nother * 3;
    const float *base = ws + (size_t)b * emd_ws_floats(n, m);
    const float *ratioL = base + n + m, *ratioR = ratioL + (size_t)kLevels * n;
    const float *rself = KSIDE ? ratioL : ratioR, *rother = KSIDE ? ratioR : ratioL;
    float xs = 0, ys = 0, zs = 0, rs[kLevels];
#pragma unroll
    for (int li = 0; li < kLevels; ++li) rs[li] = 0.f;
    if (i < nself) {
        xs = Xs[i * 3 + 0], ys = Xs[i * 3 + 1], zs = Xs[i * 3 + 2];
#pragma unroll
        for (int li = 0; li < kLevels; ++li) rs[li] = rself[(size_t)li * nself + i];
    }
    float sub = 0.f, gx = 0.f, gy = 0.f, gz = 0.f;
    for (int o0 = 0; o0 < nother; o0 += kLossTile) {
        const int oend = min(nother, o0 + kLossTile) - o0;
        __syncthreads();
        for (int o = threadIdx.x; o < oend; o += blockDim.x) {
            tpt[o] = make_float4(Xo[(o0 + o) * 3 + 0], Xo[(o0 + o) * 3 + 1], Xo[(o0 + o) * 3 + 2], 0.f);
#pragma unroll
            for (int li = 0; li < kLevels; ++li) trat[li][o] = rother[(size_t)li * nother + o0 + o];
        }
        __syncthreads();
        auto body = [&](int o) __attribute__((always_inline)) {
            const float4 t = tpt[o];
            // emd_sq(x1, x2): (x2 - x1)^2 ... -- sign-symmetric, so one expression serves both sides
            const float ex = xs - t.x, ey = ys - t.y, ez = zs - t.z;  // x_self - x_other: the gradient's direction
            const float d2 = (ex * ex + ey * ey) + ez * ez;
            float mt = 0.f;  // match[l,k], levels in order (as emd_materialize_kernel)
            const f2v dd = splat(d2);
#pragma unroll
            for (int li = 0; li < kLevels; li += 2) {
                const EmdLev L2{(f2v){emd_level(li) * kLog2eHi, emd_level(li + 1) * kLog2eHi}, (f2v){emd_level(li) * kLog2eLo, emd_level(li + 1) * kLog2eLo}};
                const f2v rl = KSIDE ? (f2v){rs[li], rs[li + 1]} : (f2v){trat[li][o], trat[li + 1][o]};
                const f2v rr = KSIDE ? (f2v){trat[li][o], trat[li + 1][o]} : (f2v){rs[li], rs[li + 1]};
                const f2v w = pk_mul(pk_mul(emd_exp2<FAST>(dd, L2), rl), rr);
                mt += w.x;
                mt += w.y;
            }
            if (KSIDE) sub += sqrtf(d2) * mt;  // cost (tf_approxmatch_g.cu:183-213)
            const float g = mt * rsqrtf(fmaxf(d2, 1e-20f));  // (:229-291)
            gx += ex * g, gy += ey * g, gz += ez * g;
        };
        int o = 0;
        for (; o + 2 <= oend; o += 2) body(o), body(o + 1);
        for (; o < oend; ++o) body(o);
    }
    if (grad && i < nself) {
        float *go = grad + ((size_t)b * nself + i) * 3;
        go[0] = gx, go[1] = gy, go[2] = gz;
    }
    if (KSIDE && partial) {
        red[threadIdx.x] = i < nself ? sub : 0.f;
        for (int s = 128; s > 0; s >>= 1) {
            __syncthreads();
            if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        }
        if (threadIdx.x == 0) partial[(size_t)b * gridDim.x + blockIdx.x] = red[0];
    }
}

// ------------------------------------------------------------------------------------------------
// The two sweeps above evaluate match[l,k] -- ten exponentials per pair -- TWICE, once per reduction axis (791 + 640 us of the
// 3.3 ms loss at B = 50, 2048 x 2048).  ONE sweep, each pair evaluated once: a workgroup owns a 64 x 64 tile of the pair matrix,
// a thread a 4 x 4 block of it (its four k-points and four l-points with their ten ratios each live in registers); the thread
// adds cost / grad1 terms of its k's over its four l's and grad2 terms of its l's over its four k's, the 16 threads that share a
// k-block (an l-block) combine through LDS in a fixed order, and the tile leaves one partial per k (cost, grad1) and per l
// (grad2): P1 [b][tiles_l][n][4], P2 [b][tiles_k][m][3].  emd_loss_reduce2d_kernel adds the tiles in ascending order:
// deterministic, no atomics.  The summation ORDER differs from emd_cost_partial_kernel / emd_grad*_kernel, so this form serves
// sn_emd_loss_fast only (the loss bar: cost 1e-5); sn_emd_loss keeps the two order-preserving sweeps (bit-identical to the
// three-call composition).
// ------------------------------------------------------------------------------------------------
constexpr int kT2 = 64;
template <bool FAST>
__global__ void __launch_bounds__(256) emd_loss_sweep2d_kernel(int n, int m, const float *__restrict__ xyz1, const float *__restrict__ xyz2,
                                                               const float *__restrict__ ws, float *__restrict__ P1, float *__restrict__ P2)
{
    __shared__ float sk[3 + kLevels][kT2], sl[3 + kLevels][kT2];
    __shared__ float red1[16][kT2][4];
    __shared__ float red2[16][kT2][3];
    const int b = blockIdx.z, kt = blockIdx.x, lt = blockIdx.y;
    const int tid = threadIdx.x;
    const float *X1 = xyz1 + (size_t)b * n * 3, *X2 = xyz2 + (size_t)b * m * 3;
    const float *base = ws + (size_t)b * emd_ws_floats(n, m);
    const float *ratioL = base + n + m, *ratioR = ratioL + (size_t)kLevels * n;
    if (tid < kT2) {
        const int k = kt * kT2 + tid;
        const bool ok = k < n;
        sk[0][tid] = ok ? X1[k * 3 + 0] : 0.f, sk[1][tid] = ok ? X1[k * 3 + 1] : 0.f, sk[2][tid] = ok ? X1[k * 3 + 2] : 0.f;
#pragma unroll
        for (int li = 0; li < kLevels; ++li) sk[3 + li][tid] = ok ? ratioL[(size_t)li * n + k] : 0.f;  // (a zero ratio: no contribution)
    } else if (tid < 2 * kT2) {
        const int t = tid - kT2, l = lt * kT2 + t;
        const bool ok = l < m;
        sl[0][t] = ok ? X2[l * 3 + 0] : 0.f, sl[1][t] = ok ? X2[l * 3 + 1] : 0.f, sl[2][t] = ok ? X2[l * 3 + 2] : 0.f;
#pragma unroll
        for (int li = 0; li < kLevels; ++li) sl[3 + li][t] = ok ? ratioR[(size_t)li * m + l] : 0.f;
    }
    __syncthreads();
    const int tk = tid & 15, tl = tid >> 4;
    float kx[4], ky[4], kz[4], kr[4][kLevels], lx[4], ly[4], lz[4], lr[4][kLevels];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        kx[a] = sk[0][tk * 4 + a], ky[a] = sk[1][tk * 4 + a], kz[a] = sk[2][tk * 4 + a];
        lx[a] = sl[0][tl * 4 + a], ly[a] = sl[1][tl * 4 + a], lz[a] = sl[2][tl * 4 + a];
#pragma unroll
        for (int li = 0; li < kLevels; ++li) kr[a][li] = sk[3 + li][tk * 4 + a], lr[a][li] = sl[3 + li][tl * 4 + a];
    }
    float cst[4], g1x[4], g1y[4], g1z[4], g2x[4], g2y[4], g2z[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) cst[a] = g1x[a] = g1y[a] = g1z[a] = g2x[a] = g2y[a] = g2z[a] = 0.f;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float ex = kx[a] - lx[c], ey = ky[a] - ly[c], ez = kz[a] - lz[c];  // x_k - x_l
            const float d2 = (ex * ex + ey * ey) + ez * ez;
            float mt = 0.f;
            const f2v dd = splat(d2);
#pragma unroll
            for (int li = 0; li < kLevels; li += 2) {
                const EmdLev L2{(f2v){emd_level(li) * kLog2eHi, emd_level(li + 1) * kLog2eHi}, (f2v){emd_level(li) * kLog2eLo, emd_level(li + 1) * kLog2eLo}};
                const f2v w = pk_mul(pk_mul(emd_exp2<FAST>(dd, L2), (f2v){kr[a][li], kr[a][li + 1]}), (f2v){lr[c][li], lr[c][li + 1]});
                mt += w.x;
                mt += w.y;
            }
            cst[a] += sqrtf(d2) * mt;
            const float g = mt * rsqrtf(fmaxf(d2, 1e-20f));
            const float tx = ex * g, ty = ey * g, tz = ez * g;
            g1x[a] += tx, g1y[a] += ty, g1z[a] += tz;
            g2x[c] -= tx, g2y[c] -= ty, g2z[c] -= tz;  // (x_l - x_k) g
        }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        *reinterpret_cast<float4 *>(&red1[tl][tk * 4 + a][0]) = make_float4(g1x[a], g1y[a], g1z[a], cst[a]);
        red2[tk][tl * 4 + a][0] = g2x[a], red2[tk][tl * 4 + a][1] = g2y[a], red2[tk][tl * 4 + a][2] = g2z[a];
    }
    __syncthreads();
    if (tid < kT2) {  // k-local tid: the 16 l-blocks in order
        float4 s4 = *reinterpret_cast<const float4 *>(&red1[0][tid][0]);
#pragma unroll
        for (int q = 1; q < 16; ++q) {
            const float4 v = *reinterpret_cast<const float4 *>(&red1[q][tid][0]);
            s4.x += v.x, s4.y += v.y, s4.z += v.z, s4.w += v.w;
        }
        const int k = kt * kT2 + tid;
        if (k < n) *reinterpret_cast<float4 *>(P1 + (((size_t)b * gridDim.y + lt) * n + k) * 4) = s4;
    } else if (tid < 2 * kT2) {  // l-local: the 16 k-blocks in order
        const int t = tid - kT2;
        float sx = red2[0][t][0], sy = red2[0][t][1], sz = red2[0][t][2];
#pragma unroll
        for (int q = 1; q < 16; ++q) sx += red2[q][t][0], sy += red2[q][t][1], sz += red2[q][t][2];
        const int l = lt * kT2 + t;
        if (l < m) {
            float *o = P2 + (((size_t)b * gridDim.x + kt) * m + l) * 3;
            o[0] = sx, o[1] = sy, o[2] = sz;
        }
    }
}

// tiles in ascending order -> grad1 (b, n, 3), grad2 (b, m, 3) (either may be NULL) and the per-workgroup cost partials that
// emd_cost_final_kernel adds up: partial [b][gridDim.x of the k side].  grid ((max(n, m) + 255) / 256, b).
__global__ void __launch_bounds__(256) emd_loss_reduce2d_kernel(int n, int m, int nkt, int nlt, const float *__restrict__ P1,
                                                                const float *__restrict__ P2, float *__restrict__ partial,
                                                                float *__restrict__ grad1, float *__restrict__ grad2)
{
    __shared__ float red[256];
    const int b = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    float c = 0.f;
    if (i < n) {
        float4 s4 = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int t = 0; t < nlt; ++t) {
            const float4 v = *reinterpret_cast<const float4 *>(P1 + (((size_t)b * nlt + t) * n + i) * 4);
            s4.x += v.x, s4.y += v.y, s4.z += v.z, s4.w += v.w;
        }
        c = s4.w;
        if (grad1) {
            float *o = grad1 + ((size_t)b * n + i) * 3;
            o[0] = s4.x, o[1] = s4.y, o[2] = s4.z;
        }
    }
    if (grad2 && i < m) {
        float sx = 0.f, sy = 0.f, sz = 0.f;
        for (int t = 0; t < nkt; ++t) {
            const float *v = P2 + (((size_t)b * nkt + t) * m + i) * 3;
            sx += v[0], sy += v[1], sz += v[2];
        }
        float *o = grad2 + ((size_t)b * m + i) * 3;
        o[0] = sx, o[1] = sy, o[2] = sz;
    }
    if (blockIdx.x < (unsigned)((n + 255) / 256)) {
        red[threadIdx.x] = c;
        for (int s2 = 128; s2 > 0; s2 >>= 1) {
            __syncthreads();
            if (threadIdx.x < s2) red[threadIdx.x] += red[threadIdx.x + s2];
        }
        if (threadIdx.x == 0) partial[(size_t)b * ((n + 255) / 256) + blockIdx.x] = red[0];
    }
}

}  // namespace sn

using namespace sn;

// ranges of the other cloud per level pass (EmdSeg): enough workgroups that the dispatcher balances the chip (>= ~2048), ranges a
// multiple of 64 points, at most one per 256 points of the other cloud
static void emd_seg_plan(int b, int nself, int nother, int &nseg, int &len)
{
    const long long base = (long long)b * ((nself + 255) / 256);
    long long s = base > 0 ? (2048 + base - 1) / base : 1;
    s = std::max<long long>(1, std::min<long long>(s, nother / 256));
    len = (int)(((nother + s - 1) / s + 63) / 64 * 64);
    nseg = (nother + len - 1) / len;
    if (nseg <= 1) nseg = 1, len = nother;
}
static long long emd_seg_floats(int b, int n, int m)
{
    int sk, lk, sl, ll;
    emd_seg_plan(b, n, m, sk, lk);
    emd_seg_plan(b, m, n, sl, ll);
    if (sk <= 1 && sl <= 1) return 0;
    const long long psum = std::max((long long)b * sk * 2 * n, (long long)b * sl * m);
    const long long ctr = (long long)b * std::max((n + 255) / 256, (m + 255) / 256);
    return psum + ((ctr + 63) / 64) * 64;
}
// per-cloud level workspace, then (behind all clouds) the segment partials and arrival counters of the level passes
long long sn_emd_workspace_floats(int b, int n, int m) { return (long long)b * (long long)emd_ws_floats(n, m) + emd_seg_floats(b, n, m); }
// floats behind the level workspace and the cost partials that the one-sweep form of sn_emd_loss_fast needs (tile partials P1, P2)
long long sn_emd_sweep2d_floats(int b, int n, int m)
{
    const long long nkt = (n + kT2 - 1) / kT2, nlt = (m + kT2 - 1) / kT2;
    return (long long)b * (nlt * n * 4 + nkt * m * 3);
}
// test / A-B hook: 0 = the level passes sweep the whole other cloud per workgroup (the one-range form of rounds 1-5), 1 (default) =
// segmented level passes (EmdSeg) where the shape makes more than one range
static int g_emd_segments = 1;
extern "C" int sn_emd_set_segments(int on)
{
    const int prev = g_emd_segments;
    g_emd_segments = on ? 1 : 0;
    return prev;
}
// test / A-B hook: 0 = sn_emd_loss_fast on the two order-preserving sweeps (as sn_emd_loss), 1 (default) = the one-sweep form
static int g_emd_sweep2d = 1;
extern "C" int sn_emd_set_sweep2d(int on)
{
    const int prev = g_emd_sweep2d;
    g_emd_sweep2d = on ? 1 : 0;
    return prev;
}

template <bool FAST>
static int emd_auction(int b, int n, int m, const float *xyz1, const float *xyz2, float *match, float *temp, sn_stream_t stream)
{
    SN_REQUIRE(b >= 0 && n >= 0 && m >= 0, "negative size");
    if (b == 0 || n == 0 || m == 0) return 0;
    SN_REQUIRE(xyz1 && xyz2 && temp, "null pointer");
    hipStream_t st = (hipStream_t)stream;
    float multiL, multiR;  // tf_approxmatch_g.cu:3-10 (integer division)
    if (n >= m)
        multiL = 1, multiR = (float)(n / m);
    else
        multiL = (float)(m / n), multiR = 1;
    EmdSeg sk{}, sl{};
    emd_seg_plan(b, n, m, sk.nseg, sk.len);
    emd_seg_plan(b, m, n, sl.nseg, sl.len);
    if (!g_emd_segments) sk.nseg = sl.nseg = 1, sk.len = m, sl.len = n;
    if (sk.nseg > 1 || sl.nseg > 1) {
        float *segbase = temp + (size_t)b * emd_ws_floats(n, m);
        const long long psum = std::max((long long)b * sk.nseg * 2 * n, (long long)b * sl.nseg * m);
        const long long nctr = (long long)b * std::max((n + 255) / 256, (m + 255) / 256);
        sk.psum = sl.psum = segbase;
        sk.counter = sl.counter = reinterpret_cast<unsigned *>(segbase + psum);
        // (the workspace is the caller's and arrives uninitialised: the counters start at zero; every launch leaves them zero)
        hipError_t e = hipMemsetAsync(sk.counter, 0, sizeof(unsigned) * (size_t)nctr, st);
        if (e != hipSuccess) return sn_set_error((int)e, "emd: %s", hipGetErrorString(e));
    }
    const dim3 gk((n + 255) / 256, b, sk.nseg), gl((m + 255) / 256, b, sl.nseg);
    for (int li = 0; li < kLevels; ++li) {
        hipLaunchKernelGGL(emd_pass_k_kernel<FAST>, gk, dim3(256), 0, st, n, m, li, xyz1, xyz2, temp, multiL, multiR, sk);
        hipLaunchKernelGGL(emd_pass_l_kernel<FAST>, gl, dim3(256), 0, st, n, m, li, xyz1, xyz2, temp, sl);
    }
    if (match)
        hipLaunchKernelGGL(emd_materialize_kernel, dim3((n + 255) / 256, (m + 15) / 16, b), dim3(256), 0, st, n, m,
                           xyz1, xyz2, temp, match);
    SN_LAUNCH_CHECK();
    return 0;
}

extern "C" int sn_approxmatch(int b, int n, int m, const float *xyz1, const float *xyz2, float *match, float *temp,
                              sn_stream_t stream)
{
    return emd_auction<false>(b, n, m, xyz1, xyz2, match, temp, stream);
}

extern "C" int sn_matchcost(int b, int n, int m, const float *xyz1, const float *xyz2, const float *match,
                            float *cost, float *workspace, sn_stream_t stream)
{
    SN_REQUIRE(b >= 0 && n >= 0 && m >= 0, "negative size");
    if (b == 0) return 0;
    SN_REQUIRE(cost, "null pointer");
    hipStream_t st = (hipStream_t)stream;
    if (n == 0 || m == 0) {
        hipError_t e = hipMemsetAsync(cost, 0, sizeof(float) * b, st);
        return e == hipSuccess ? 0 : sn_set_error((int)e, "sn_matchcost: %s", hipGetErrorString(e));
    }
    SN_REQUIRE(xyz1 && xyz2 && match && workspace, "null pointer");
    const int nparts = (n + 255) / 256;  // workspace: b * nparts floats (sn_workspace_bytes("matchcost", ...))
    hipLaunchKernelGGL(emd_cost_partial_kernel, dim3(nparts, b), dim3(256), 0, st, n, m, xyz1, xyz2, match, workspace);
    hipLaunchKernelGGL(emd_cost_final_kernel, dim3((b + 255) / 256), dim3(256), 0, st, b, nparts, workspace, cost);
    SN_LAUNCH_CHECK();
    return 0;
}

extern "C" int sn_matchcost_grad(int b, int n, int m, const float *xyz1, const float *xyz2, const float *match,
                                 float *grad1, float *grad2, sn_stream_t stream)
{
    SN_REQUIRE(b >= 0 && n >= 0 && m >= 0, "negative size");
    if (b == 0 || n == 0 || m == 0) return 0;
    SN_REQUIRE(xyz1 && xyz2 && match, "null pointer");
    hipStream_t st = (hipStream_t)stream;
    if (grad1) hipLaunchKernelGGL(emd_grad1_kernel, dim3((n + 255) / 256, b), dim3(256), 0, st, n, m, xyz1, xyz2, match, grad1);
    if (grad2) hipLaunchKernelGGL(emd_grad2_kernel, dim3((m + 3) / 4, b), dim3(256), 0, st, n, m, xyz1, xyz2, match, grad2);
    SN_LAUNCH_CHECK();
    return 0;
}

// cost (b) = match_cost(approx_match(xyz1, xyz2)) and its gradients without materialising match (see emd_loss_sweep_kernel).
// temp: sn_workspace_bytes("emd_loss", b, n, m, 0) bytes.  grad1 / grad2 may be NULL.
template <bool FAST>
static int emd_loss_impl(int b, int n, int m, const float *xyz1, const float *xyz2, float *cost, float *grad1, float *grad2,
                         float *temp, sn_stream_t stream, const char *who)
{
    if (!(b >= 0 && n >= 0 && m >= 0)) return sn_set_error(SN_ERR_BAD_ARGUMENT, "%s: negative size", who);
    if (b == 0) return 0;
    if (!cost) return sn_set_error(SN_ERR_BAD_ARGUMENT, "%s: null pointer", who);
    hipStream_t st = (hipStream_t)stream;
    if (n == 0 || m == 0) {
        hipError_t e = hipMemsetAsync(cost, 0, sizeof(float) * b, st);
        return e == hipSuccess ? 0 : sn_set_error((int)e, "%s: %s", who, hipGetErrorString(e));
    }
    if (!(xyz1 && xyz2 && temp)) return sn_set_error(SN_ERR_BAD_ARGUMENT, "%s: null pointer", who);
    int rc = emd_auction<FAST>(b, n, m, xyz1, xyz2, nullptr, temp, stream);  // the 20 level passes; no materialisation
    if (rc) return rc;
    float *partial = temp + sn_emd_workspace_floats(b, n, m);
    const int nparts = (n + 255) / 256;
    if (FAST && g_emd_sweep2d) {
        // the default loss form: every pair's match value evaluated once for cost, grad1 and grad2 (see emd_loss_sweep2d_kernel)
        const int nkt = (n + kT2 - 1) / kT2, nlt = (m + kT2 - 1) / kT2;
        float *P1 = partial + (size_t)b * nparts, *P2 = P1 + (size_t)b * nlt * n * 4;
        hipLaunchKernelGGL((emd_loss_sweep2d_kernel<FAST>), dim3(nkt, nlt, b), dim3(256), 0, st, n, m, xyz1, xyz2, temp, P1, P2);
        hipLaunchKernelGGL(emd_loss_reduce2d_kernel, dim3((std::max(n, m) + 255) / 256, b), dim3(256), 0, st, n, m, nkt, nlt, P1, P2, partial,
                           grad1, grad2);
        hipLaunchKernelGGL(emd_cost_final_kernel, dim3((b + 255) / 256), dim3(256), 0, st, b, nparts, partial, cost);
        SN_LAUNCH_CHECK();
        return 0;
    }
    hipLaunchKernelGGL((emd_loss_sweep_kernel<true, FAST>), dim3(nparts, b), dim3(256), 0, st, n, m, xyz1, xyz2, temp, partial, grad1);
    hipLaunchKernelGGL(emd_cost_final_kernel, dim3((b + 255) / 256), dim3(256), 0, st, b, nparts, partial, cost);
    if (grad2)
        hipLaunchKernelGGL((emd_loss_sweep_kernel<false, FAST>), dim3((m + 255) / 256, b), dim3(256), 0, st, n, m, xyz1, xyz2, temp,
                           (float *)nullptr, grad2);
    SN_LAUNCH_CHECK();
    return 0;
}

extern "C" int sn_emd_loss(int b, int n, int m, const float *xyz1, const float *xyz2, float *cost, float *grad1, float *grad2,
                           float *temp, sn_stream_t stream)
{
    (void)hipGetLastError();
    return emd_loss_impl<false>(b, n, m, xyz1, xyz2, cost, grad1, grad2, temp, stream, "sn_emd_loss");
}

// The same loss with the reference op's own exponential, __expf = v_exp_f32(x log2 e) (tf_approxmatch_g.cu:52,97,151), in the
// auction and in the sweeps: the form a training loss wants (1e-5 on the cost); NOT bit-identical to the three-call
// composition on sn_approxmatch's match, which keeps the compensated exponential for its per-entry bar.
extern "C" int sn_emd_loss_fast(int b, int n, int m, const float *xyz1, const float *xyz2, float *cost, float *grad1, float *grad2,
                                float *temp, sn_stream_t stream)
{
    (void)hipGetLastError();
    return emd_loss_impl<true>(b, n, m, xyz1, xyz2, cost, grad1, grad2, temp, stream, "sn_emd_loss_fast");
}
