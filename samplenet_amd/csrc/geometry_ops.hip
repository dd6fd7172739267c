// geometry_ops.hip -- backward kernels and the small gather/scatter ops of the hot path (gfx950).
//
//   chamfer_backward      deterministic, sequential-order accumulation (bit-exact with the CPU
//                         reference loops chamfer_distance.cpp:114-177; the reference GPU path uses
//                         unordered float atomics, chamfer_distance.cu:158-187)
//   soft_project_backward fused backward of SoftProjection.project (soft_projection.py:138-152)
//   soft_weights_*, weighted_gather_*   the same math split at the weights, for the
//                         propagate / project_and_propagate actions (soft_projection.py:101-136)
//   group_point*, grouping_operation*   tf_grouping_g.cu:40-78 and the pointnet2 layout twin
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "sn_common.h"
#include "step_tail.h"

#pragma clang fp contract(off)  // reference arithmetic is product-then-sum (no FMA)

namespace sn {

// ------------------------------------------------------------------------------------------------
// Chamfer backward.  For the target set T (nt points) against the source set S (ns points):
//   grad_T[j] = 2 gT[j] (t_j - s_{idxT[j]})  -  sum_{l : idxS[l] == j} 2 gS[l] (s_l - t_j)
// own_first selects where the 1:1 term enters the sequential sum, to mirror the reference order:
//   xyz1 side: own term first, then scatter terms in ascending l   (chamfer_distance.cpp:140-155 then :169-171)
//   xyz2 side: scatter terms in ascending j first, then own term    (:152-154 then :166-168)
// A wave owns one target; its lanes scan 64 sources per step, a ballot yields the matching
// sources and they are accumulated in ascending order (wave-uniform arithmetic).
// ------------------------------------------------------------------------------------------------
// Implicit upstream gradients (fused simplification loss, samplenet.py:171-181): with gL = d loss / d (scalar loss),
//   g(dist of target j)  = gL * (ct + (j == argmax_t[b] ? cmax_t : 0)),   g(dist of source l) = gL * (cs + (l == argmax_s[b] ? cmax_s : 0))
// so no per-point gradient tensors are materialised.  gL == NULL: explicit gT / gS arrays are used.
struct sn_xyz3 {
    float x, y, z;
};

constexpr int kMaxPrefixes = 16;  // nested prefixes / grouped evaluations a launch takes by value

struct ImplicitGrad {
    const float *gL;
    const int *argmax_t, *argmax_s;
    float ct, cmax_t, cs, cmax_s;
    float gscale = 1.f;  // upstream gradient = *gL * gscale (rounded once, like a separate scaling op in front would)
};

__global__ void __launch_bounds__(256) chamfer_bwd_kernel(int nt, int ns, const float *__restrict__ T,
                                                          const float *__restrict__ S,
                                                          const float *__restrict__ gT,
                                                          const int *__restrict__ idxT,
                                                          const float *__restrict__ gS,
                                                          const int *__restrict__ idxS, float *__restrict__ gradT,
                                                          int own_first, ImplicitGrad ig, int t_layout)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwaves = blockDim.x >> 6;
    const int b = blockIdx.x;
    T += (size_t)b * nt * 3, S += (size_t)b * ns * 3;
    idxT += (size_t)b * nt, idxS += (size_t)b * ns;
    const bool implicit = ig.gL != nullptr;
    float gLv = 0.f;
    int amt = -1, ams = -1;
    if (implicit) {
        gLv = *ig.gL * ig.gscale;
        if (ig.argmax_t) amt = ig.argmax_t[b];
        if (ig.argmax_s) ams = ig.argmax_s[b];
    } else {
        gT += (size_t)b * nt, gS += (size_t)b * ns;
    }
    gradT += (size_t)b * nt * 3;
    for (int j = blockIdx.y * nwaves + wave; j < nt; j += gridDim.y * nwaves) {
        // target coordinates / gradient: point-major (nt,3) or channel-major (3,nt) -- the sampler's FC head emits (3,M)
        const int o0 = t_layout ? j : j * 3, os = t_layout ? nt : 1;
        const float tx = T[o0], ty = T[o0 + os], tz = T[o0 + 2 * os];
        const int j2 = idxT[j];
        const float g = (implicit ? gLv * (ig.ct + (j == amt ? ig.cmax_t : 0.f)) : gT[j]) * 2;
        const float ox = g * (tx - S[j2 * 3 + 0]);
        const float oy = g * (ty - S[j2 * 3 + 1]);
        const float oz = g * (tz - S[j2 * 3 + 2]);
        float ax = 0.f, ay = 0.f, az = 0.f;
        if (own_first) ax += ox, ay += oy, az += oz;
        for (int l0 = 0; l0 < ns; l0 += 64 * 8) {
            int is[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {  // 8 independent index loads in flight
                const int l = l0 + q * 64 + lane;
                is[q] = idxS[l < ns ? l : 0];
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int l = l0 + q * 64 + lane;
                sn_u64 mask = __ballot(l < ns && is[q] == j);
                while (mask) {  // matching sources in ascending index order
                    const int ll = l0 + q * 64 + __builtin_ctzll(mask);
                    mask &= mask - 1;
                    const float gg = (implicit ? gLv * (ig.cs + (ll == ams ? ig.cmax_s : 0.f)) : gS[ll]) * 2;
                    ax -= gg * (S[ll * 3 + 0] - tx);
                    ay -= gg * (S[ll * 3 + 1] - ty);
                    az -= gg * (S[ll * 3 + 2] - tz);
                }
            }
        }
        if (!own_first) ax += ox, ay += oy, az += oz;
        if (lane < 3) gradT[o0 + lane * os] = lane == 0 ? ax : (lane == 1 ? ay : az);
    }
}

// Register-resident variant for ns <= 64*PPL (the sampler's 1024-point clouds): every lane loads its PPL sources
// (coordinates, nearest-target index, upstream gradient) ONCE, all loads in flight together; per target the matching
// lanes form their contribution in parallel and only the ordered accumulation (ascending source index, as the
// reference CPU loop) is sequential -- on registers via v_readlane, no memory latency inside it.
template <int PPL>
__global__ void __launch_bounds__(256) chamfer_bwd_reg_kernel(int nt, int ns, const float *__restrict__ T,
                                                              const float *__restrict__ S,
                                                              const float *__restrict__ gT,
                                                              const int *__restrict__ idxT,
                                                              const float *__restrict__ gS,
                                                              const int *__restrict__ idxS, float *__restrict__ gradT,
                                                              int own_first, ImplicitGrad ig, int t_layout)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwaves = blockDim.x >> 6;
    const int b = blockIdx.x;
    T += (size_t)b * nt * 3, S += (size_t)b * ns * 3;
    idxT += (size_t)b * nt, idxS += (size_t)b * ns;
    const bool implicit = ig.gL != nullptr;
    float gLv = 0.f;
    int amt = -1, ams = -1;
    if (implicit) {
        gLv = *ig.gL * ig.gscale;
        if (ig.argmax_t) amt = ig.argmax_t[b];
        if (ig.argmax_s) ams = ig.argmax_s[b];
    } else {
        gT += (size_t)b * nt, gS += (size_t)b * ns;
    }
    gradT += (size_t)b * nt * 3;

    float sx[PPL], sy[PPL], sz[PPL], gg[PPL];
    int is[PPL];
#pragma unroll
    for (int i = 0; i < PPL; ++i) {
        const int l = i * 64 + lane;
        const int lc = l < ns ? l : 0;
        const sn_xyz3 sv = *reinterpret_cast<const sn_xyz3 *>(S + (size_t)lc * 3);  // one 12-byte load per source point
        sx[i] = sv.x, sy[i] = sv.y, sz[i] = sv.z;
        is[i] = l < ns ? idxS[lc] : -1;
        gg[i] = (implicit ? gLv * (ig.cs + (l == ams ? ig.cmax_s : 0.f)) : gS[lc]) * 2;
    }
    for (int j = blockIdx.y * nwaves + wave; j < nt; j += gridDim.y * nwaves) {
        // target coordinates / gradient: point-major (nt,3) or channel-major (3,nt) -- the sampler's FC head emits (3,M)
        const int o0 = t_layout ? j : j * 3, os = t_layout ? nt : 1;
        const float tx = T[o0], ty = T[o0 + os], tz = T[o0 + 2 * os];
        const int j2 = idxT[j];
        const float g = (implicit ? gLv * (ig.ct + (j == amt ? ig.cmax_t : 0.f)) : gT[j]) * 2;
        const float ox = g * (tx - S[j2 * 3 + 0]);
        const float oy = g * (ty - S[j2 * 3 + 1]);
        const float oz = g * (tz - S[j2 * 3 + 2]);
        float ax = 0.f, ay = 0.f, az = 0.f;
        if (own_first) ax += ox, ay += oy, az += oz;
#pragma unroll
        for (int i = 0; i < PPL; ++i) {
            sn_u64 mask = __ballot(is[i] == j);
            if (mask) {
                const float cx = gg[i] * (sx[i] - tx), cy = gg[i] * (sy[i] - ty), cz = gg[i] * (sz[i] - tz);
                while (mask) {  // ascending source index: lanes of slot i in lane order, slots in order
                    const int t = __builtin_ctzll(mask);
                    mask &= mask - 1;
                    ax -= readlane_f(cx, t);
                    ay -= readlane_f(cy, t);
                    az -= readlane_f(cz, t);
                }
            }
        }
        if (!own_first) ax += ox, ay += oy, az += oz;
        if (lane < 3) gradT[o0 + lane * os] = lane == 0 ? ax : (lane == 1 ? ay : az);
    }
}

// workgroups (of 4 waves) the targets of all clouds are spread over: every wave loads the whole source set into registers
// before its first target, so fewer, longer-lived waves amortise that load (1024 groups = one target per wave at the
// sampler's sizes: measured 9 us; swept 64..1024 at B = 32: 1024 is fastest)
static const int kChamferBwdGroups = 1024;

static void launch_chamfer_bwd(int b, int ysplit, int nt, int ns, const float *T, const float *S, const float *gT,
                               const int *idxT, const float *gS, const int *idxS, float *gradT, int own_first,
                               const ImplicitGrad &ig, hipStream_t st, int t_layout = 0)
{
    const dim3 grid(b, ysplit), block(256);
#define SN_CB(PPL_) \
    hipLaunchKernelGGL(chamfer_bwd_reg_kernel<PPL_>, grid, block, 0, st, nt, ns, T, S, gT, idxT, gS, idxS, gradT, own_first, ig, t_layout)
    if (ns <= 64) SN_CB(1);
    else if (ns <= 256) SN_CB(4);
    else if (ns <= 1024) SN_CB(16);
    else if (ns <= 2048) SN_CB(32);
    else
        hipLaunchKernelGGL(chamfer_bwd_kernel, grid, block, 0, st, nt, ns, T, S, gT, idxT, gS, idxS, gradT, own_first, ig, t_layout);
#undef SN_CB
}

// ------------------------------------------------------------------------------------------------
// SoftProjection backward (fused 'project').  One wave per query, lane t < K = neighbour t.
// ------------------------------------------------------------------------------------------------
struct SoftBwdArgs {
    const float *P;
    const float *Q;
    const int *idx;
    const float *temperature;
    float min_sigma;
    int p_layout, q_layout;
    int n, m, k;
    const float *grad_proj;  // fused mode
    int gproj_layout;
    const float *weights_in;    // split mode: saved weights (unused, recomputed) -- kept for ABI symmetry
    const float *grad_weights;  // split mode
    float *grad_Q;
    int gq_layout;
    float *grad_P;  // (b,3,n) channel-major in split mode, p_layout in fused mode; atomics
    // ordered route (sn_*_backward_ordered): the per-(query, neighbour) contributions to grad_P are STORED here instead of added
    // with atomics -- entry e = j * k + t; (b, m k, 3) for a point-major grad_P, (b, 3, m k) for a channel-major one -- and summed
    // by index_add_ordered_kernel in ascending entry order: the reference's CPU loop, deterministic
    float *gp_contrib;
    float *grad_sigma_partial;
    // fused mode, grad_proj == NULL: the upstream gradient is the same for every element, *gconst / gconst_div
    // (the sampler step's mean(proj) term); accumulate_q: add to grad_Q instead of overwriting it
    const float *gconst;
    float gconst_div;
    int accumulate_q;
};

// One query of the soft-projection backward, executed by a whole wave: returns the gradient to the query (aq*,
// wave-uniform) and its share of d loss / d sigma; accumulates grad_P if requested.  Three steps, so that a caller can
// request the query's memory operands ahead of other work: the neighbour indices, the neighbours' coordinates, the math.
//   K <= 16: every row of 16 lanes holds the K neighbours (lane t of a row = neighbour t) and the ascending-k sums run
//            as sequential DPP scans inside the rows (lane t adds its term to lane t-1's running sum -- the order of the
//            loops below), the four closing sums (gradient x, y, z and the sigma term) one per row;
//   else   : lane t < K = neighbour t, sums by v_readlane loops.
struct SoftQuery {
    float qx, qy, qz;  // the query
    int id;            // this lane's neighbour
    float gx, gy, gz;  // its coordinates
    float go0, go1, go2;  // fused mode with an explicit upstream gradient: d loss / d proj of this query
};

__device__ __forceinline__ float dpp_row_shr1(float x)  // lane below within the row of 16; 0 into the row's lane 0
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x111, 0xf, 0xf, true));
}
__device__ __forceinline__ float dpp_row_shr1_self(float x)  // ... the row's lane 0 reads itself
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(x), __float_as_int(x), 0x111, 0xf, 0xf, false));
}

__device__ __forceinline__ void soft_bwd_load_index(const SoftBwdArgs &a, int b, int j, int lane, const float *__restrict__ Qb,
                                                    SoftQuery &q)
{
    const int m = a.m, K = a.k;
    q.qx = Qb[pt_off(a.q_layout, m, j, 0)];
    q.qy = Qb[pt_off(a.q_layout, m, j, 1)];
    q.qz = Qb[pt_off(a.q_layout, m, j, 2)];
    const int sub = K <= 16 ? (lane & 15) : lane;
    q.id = sub < K ? a.idx[((size_t)b * m + j) * K + sub] : 0;
    q.go0 = q.go1 = q.go2 = 0.f;
    if (a.grad_proj) {  // (requested here, with the query's other first-round loads, not in front of its use)
        const float *gp = a.grad_proj + (size_t)b * 3 * m;
        q.go0 = gp[pt_off(a.gproj_layout, m, j, 0)];
        q.go1 = gp[pt_off(a.gproj_layout, m, j, 1)];
        q.go2 = gp[pt_off(a.gproj_layout, m, j, 2)];
    }
}

__device__ __forceinline__ void soft_bwd_load_points(const SoftBwdArgs &a, int lane, const float *__restrict__ Pb, SoftQuery &q)
{
    const int n = a.n, K = a.k;
    const int sub = K <= 16 ? (lane & 15) : lane;
    q.gx = q.gy = q.gz = 0.f;
    if (sub < K) {
        q.gx = Pb[pt_off(a.p_layout, n, q.id, 0)];
        q.gy = Pb[pt_off(a.p_layout, n, q.id, 1)];
        q.gz = Pb[pt_off(a.p_layout, n, q.id, 2)];
    }
}

template <bool FUSED>
__device__ __forceinline__ void soft_bwd_math(const SoftBwdArgs &a, int b, int j, int lane, float sigma, const SoftQuery &q,
                                              float &aqx_o, float &aqy_o, float &aqz_o, float &asg_o)
{
    const int n = a.n, m = a.m, K = a.k;
    const bool rows = K <= 16;
    const int sub = rows ? (lane & 15) : lane;
    const bool act = sub < K;
    const int id = q.id;
    const float gx = q.gx, gy = q.gy, gz = q.gz;
    const float dx = gx - q.qx, dy = gy - q.qy, dz = gz - q.qz;
    const float d = (dx * dx + dy * dy) + dz * dz;
    const float s = act ? -(d / sigma) : -INFINITY;
    float mx, den;
    if (rows) {
        mx = s;
        for (int t = 1; t < K; ++t) mx = fmaxf(s, dpp_row_shr1_self(mx));
        mx = readlane_f(mx, K - 1);
    } else {
        mx = readlane_f(s, 0);  // neighbours are stored ascending in distance; the scan below only matters
        for (int t = 1; t < K; ++t) mx = fmaxf(mx, readlane_f(s, t));  // if a caller passes unsorted indices
    }
    const float e = act ? expf(s - mx) : 0.f;
    if (rows) {
        den = e;
        for (int t = 1; t < K; ++t) den = e + dpp_row_shr1(den);
        den = readlane_f(den, K - 1);
    } else {
        den = 0.f;
        for (int t = 0; t < K; ++t) den += readlane_f(e, t);
    }
    const float w = e / den;

    float gw;  // d loss / d w_t
    float go0 = 0.f, go1 = 0.f, go2 = 0.f;
    if (FUSED) {
        if (a.grad_proj) {
            go0 = q.go0, go1 = q.go1, go2 = q.go2;
        } else {
            go0 = go1 = go2 = *a.gconst / a.gconst_div;
        }
        gw = (go0 * gx + go1 * gy) + go2 * gz;
    } else {
        gw = act ? a.grad_weights[((size_t)b * m + j) * K + sub] : 0.f;
    }
    float dot;
    if (rows) {
        const float wg = w * gw;
        dot = wg;
        for (int t = 1; t < K; ++t) dot = wg + dpp_row_shr1(dot);
        dot = readlane_f(dot, K - 1);
    } else {
        dot = 0.f;
        for (int t = 0; t < K; ++t) dot += readlane_f(w, t) * readlane_f(gw, t);
    }
    const float gs = act ? w * (gw - dot) : 0.f;  // softmax backward
    const float gd = -gs / sigma;                 // s = -d / sigma
    const float cx = 2.0f * gd * dx, cy = 2.0f * gd * dy, cz = 2.0f * gd * dz;
    const float sg = act ? gs * d / (sigma * sigma) : 0.f;
    float aqx = 0.f, aqy = 0.f, aqz = 0.f, asg = 0.f;
    if (rows) {
        // row 0: 0 - cx_0 - cx_1 ..., row 1: cy, row 2: cz, row 3: 0 + sg_0 + sg_1 ...
        const int row = lane >> 4;
        const float v = row == 0 ? -cx : (row == 1 ? -cy : (row == 2 ? -cz : sg));
        float acc = v;
        for (int t = 1; t < K; ++t) acc = v + dpp_row_shr1(acc);
        aqx = readlane_f(acc, K - 1), aqy = readlane_f(acc, 16 + K - 1), aqz = readlane_f(acc, 32 + K - 1);
        asg = readlane_f(acc, 48 + K - 1);
    } else {
        for (int t = 0; t < K; ++t) {
            aqx -= readlane_f(cx, t);
            aqy -= readlane_f(cy, t);
            aqz -= readlane_f(cz, t);
            asg += readlane_f(sg, t);
        }
    }
    if (a.gp_contrib && lane < K) {
        const int lay = FUSED ? a.p_layout : SN_LAYOUT_BCN;
        const float ex = FUSED ? go0 * w : 0.f, ey = FUSED ? go1 * w : 0.f, ez = FUSED ? go2 * w : 0.f;
        const size_t ne = (size_t)m * K, e = (size_t)j * K + lane;
        float *cb = a.gp_contrib + (size_t)b * ne * 3;
        if (lay == SN_LAYOUT_BNC) cb[e * 3] = ex + cx, cb[e * 3 + 1] = ey + cy, cb[e * 3 + 2] = ez + cz;
        else cb[e] = ex + cx, cb[ne + e] = ey + cy, cb[2 * ne + e] = ez + cz;
    } else if (a.grad_P && lane < K) {
        float *gpb = a.grad_P + (size_t)b * 3 * n;
        const int lay = FUSED ? a.p_layout : SN_LAYOUT_BCN;
        const float ex = FUSED ? go0 * w : 0.f, ey = FUSED ? go1 * w : 0.f, ez = FUSED ? go2 * w : 0.f;
        atomicAdd(&gpb[pt_off(lay, n, id, 0)], ex + cx);
        atomicAdd(&gpb[pt_off(lay, n, id, 1)], ey + cy);
        atomicAdd(&gpb[pt_off(lay, n, id, 2)], ez + cz);
    }
    aqx_o = aqx, aqy_o = aqy, aqz_o = aqz, asg_o = asg;
}

template <bool FUSED>
__device__ __forceinline__ void soft_bwd_query(const SoftBwdArgs &a, int b, int j, int lane, float sigma,
                                               const float *__restrict__ Pb, const float *__restrict__ Qb, float &aqx_o,
                                               float &aqy_o, float &aqz_o, float &asg_o)
{
    SoftQuery q;
    soft_bwd_load_index(a, b, j, lane, Qb, q);
    soft_bwd_load_points(a, lane, Pb, q);
    soft_bwd_math<FUSED>(a, b, j, lane, sigma, q, aqx_o, aqy_o, aqz_o, asg_o);
}

template <bool FUSED>
__global__ void __launch_bounds__(256) soft_bwd_kernel(SoftBwdArgs a)
{
    __shared__ float s_part[4];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwaves = blockDim.x >> 6;
    const int b = blockIdx.x;
    const int n = a.n, m = a.m;
    const float *__restrict__ Pb = a.P + (size_t)b * 3 * n;
    const float *__restrict__ Qb = a.Q + (size_t)b * 3 * m;
    const float T = *a.temperature;
    const float sigma = sn_sigma(T, a.min_sigma);
    float gsig = 0.f;  // this wave's share of d loss / d sigma

    for (int j = blockIdx.y * nwaves + wave; j < m; j += gridDim.y * nwaves) {
        float aqx, aqy, aqz, asg;
        soft_bwd_query<FUSED>(a, b, j, lane, sigma, Pb, Qb, aqx, aqy, aqz, asg);
        gsig += asg;
        if (a.grad_Q && lane < 3) {
            const float o = lane == 0 ? aqx : (lane == 1 ? aqy : aqz);
            float *dst = a.grad_Q + (size_t)b * 3 * m + pt_off(a.gq_layout, m, j, lane);
            *dst = a.accumulate_q ? *dst + o : o;
        }
    }
    // fixed-order block reduction of the sigma gradient: wave 0..3 in order
    if (lane == 0) s_part[wave] = gsig;
    __syncthreads();
    if (threadIdx.x == 0 && a.grad_sigma_partial) {
        float tot = 0.f;
        for (int w2 = 0; w2 < nwaves; ++w2) tot += s_part[w2];
        a.grad_sigma_partial[(size_t)b * gridDim.y + blockIdx.y] = tot;
    }
}

// softmax weights alone (split path)
__global__ void __launch_bounds__(256) soft_weights_fwd_kernel(int n, int m, int K, const float *__restrict__ P,
                                                               const float *__restrict__ Q,
                                                               const int *__restrict__ idx,
                                                               const float *__restrict__ temperature,
                                                               float min_sigma, float *__restrict__ weights)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwaves = blockDim.x >> 6;
    const int b = blockIdx.x;
    const float *Pb = P + (size_t)b * 3 * n, *Qb = Q + (size_t)b * 3 * m;
    const float T = *temperature;
    const float sigma = sn_sigma(T, min_sigma);
    for (int j = blockIdx.y * nwaves + wave; j < m; j += gridDim.y * nwaves) {
        const bool act = lane < K;
        const int id = act ? idx[((size_t)b * m + j) * K + lane] : 0;
        const float dx = (act ? Pb[id] : 0.f) - Qb[j];
        const float dy = (act ? Pb[n + id] : 0.f) - Qb[m + j];
        const float dz = (act ? Pb[2 * n + id] : 0.f) - Qb[2 * m + j];
        const float d = (dx * dx + dy * dy) + dz * dz;
        const float s = act ? -(d / sigma) : -INFINITY;
        float mx = s;
#pragma unroll
        for (int t = 1; t < 64; t <<= 1) mx = fmaxf(mx, __shfl_xor(mx, t));
        const float e = act ? expf(s - mx) : 0.f;
        float den = 0.f;
        for (int t = 0; t < K; ++t) den += readlane_f(e, t);
        if (act) weights[((size_t)b * m + j) * K + lane] = e / den;
    }
}

// out[b,c,j] = sum_t w[b,j,t] * X[b,c,idx[b,j,t]]   (ascending t, product then sum)
__global__ void __launch_bounds__(256) weighted_gather_fwd_kernel(int c, int n, int m, int K,
                                                                  const float *__restrict__ X,
                                                                  const int *__restrict__ idx,
                                                                  const float *__restrict__ w,
                                                                  float *__restrict__ out)
{
    const int b = blockIdx.y;
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // over c*m
    if (e >= (size_t)c * m) return;
    const int ch = (int)(e / m), j = (int)(e % m);
    const float *Xc = X + ((size_t)b * c + ch) * n;
    const int *id = idx + ((size_t)b * m + j) * K;
    const float *wj = w + ((size_t)b * m + j) * K;
    float acc = 0.f;
    for (int t = 0; t < K; ++t) acc += Xc[id[t]] * wj[t];
    out[((size_t)b * c + ch) * m + j] = acc;
}

// grad_w[b,j,t] = sum_c go[b,c,j] X[b,c,idx];  grad_X[b,c,idx] += go[b,c,j] w[b,j,t]  (atomics)
__global__ void __launch_bounds__(256) weighted_gather_bwd_kernel(int c, int n, int m, int K,
                                                                  const float *__restrict__ X,
                                                                  const int *__restrict__ idx,
                                                                  const float *__restrict__ w,
                                                                  const float *__restrict__ go,
                                                                  float *__restrict__ gw, float *__restrict__ gX,
                                                                  float *__restrict__ contrib = nullptr)
{
    const int b = blockIdx.y;
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // over m*K
    if (e >= (size_t)m * K) return;
    const int j = (int)(e / K);
    const int id = idx[(size_t)b * m * K + e];
    const float wt = w[(size_t)b * m * K + e];
    float acc = 0.f;
    for (int ch = 0; ch < c; ++ch) {
        const float g = go[((size_t)b * c + ch) * m + j];
        acc += g * X[((size_t)b * c + ch) * n + id];
        if (contrib) contrib[((size_t)b * c + ch) * m * K + e] = g * wt;  // (b, c, m k): summed in entry order afterwards
        else if (gX) atomicAdd(&gX[((size_t)b * c + ch) * n + id], g * wt);
    }
    if (gw) gw[(size_t)b * m * K + e] = acc;
}

// ------------------------------------------------------------------------------------------------
// group_point (TF layout) and grouping_operation (channel-major layout)
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) group_point_kernel(int n, int c, int m, int ns, const float *__restrict__ points,
                                                          const int *__restrict__ idx, float *__restrict__ out)
{
    const int b = blockIdx.y;
    const size_t tot = (size_t)m * ns * c;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += (size_t)gridDim.x * blockDim.x) {
        const int l = (int)(e % c);
        const size_t jk = e / c;
        const int ii = idx[(size_t)b * m * ns + jk];
        out[(size_t)b * tot + e] = points[((size_t)b * n + ii) * c + l];
    }
}

// Gradient of the two gathers: dst row r = sum of the src entries e whose index names r, added in ASCENDING entry order -- the
// order of the reference's CPU loops (tf_grouping.cpp's group_point_grad_cpu; its GPU kernels, tf_grouping_g.cu:60-78, use
// unordered float atomics), hence deterministic and bit-identical to the oracle.  A thread owns one destination row and walks
// the cloud's ne = m * nsample indices adding the rare hits; no atomics, no memset (every row is written).  Indices outside
// [0, n) name no row and are ignored.
//   CHANNEL_MAJOR = false: src (ne, c) / dst (n, c)  [group_point];  true: src (c, ne) / dst (c, n)  [grouping_operation]
template <bool CHANNEL_MAJOR>
__global__ void __launch_bounds__(256) index_add_ordered_kernel(int n, int c, int ne, const int *__restrict__ idx,
                                                                const float *__restrict__ src, float *__restrict__ dst)
{
    const int b = blockIdx.y, lane = threadIdx.x & 63;
    const int row0 = blockIdx.x * 256 + (threadIdx.x & ~63);  // this wave's 64 rows
    const int row = row0 + lane;
    const bool live = row < n;
    idx += (size_t)b * ne;
    src += (size_t)b * ne * c;
    dst += (size_t)b * n * c;
    auto d = [&](int l) -> float & { return dst[CHANNEL_MAJOR ? (size_t)l * n + row : (size_t)row * c + l]; };
    if (live)
        for (int l = 0; l < c; ++l) d(l) = 0.f;
    // 64 indices per trip (one coalesced load); the wave then visits, in ascending order, only the entries that name one of
    // ITS rows (on average one in n / 64), each handled by the lane that owns the row
    for (int e0 = 0; e0 < ne; e0 += 64) {
        const int mine = e0 + lane < ne ? idx[e0 + lane] : -1;
        sn_u64 hits = __ballot((unsigned)(mine - row0) < 64u);
        while (hits) {
            const int t = __builtin_ctzll(hits);
            hits &= hits - 1;
            const int id = __builtin_amdgcn_readlane(mine, t), e = e0 + t;
            if (live && id == row)
                for (int l = 0; l < c; ++l) d(l) += src[CHANNEL_MAJOR ? (size_t)l * ne + e : (size_t)e * c + l];
        }
    }
}

// Large gathers (PointNet++-sized grouping: n and m * nsample both in the tens of thousands): the ordered scan above costs
// (n / 64) * ne index loads per cloud, so beyond kIndexAddOrderedWork the gradient is scattered with float atomics into a
// zeroed destination -- what the reference's own GPU kernels do (tf_grouping_g.cu:60-78): same sums, unordered additions.
constexpr long long kIndexAddOrderedWork = 1ll << 26;
template <bool CHANNEL_MAJOR>
__global__ void __launch_bounds__(256) index_add_atomic_kernel(int n, int c, long long ne, const int *__restrict__ idx,
                                                               const float *__restrict__ src, float *__restrict__ dst)
{
    const int b = blockIdx.y;
    idx += (size_t)b * ne;
    src += (size_t)b * ne * c;
    dst += (size_t)b * n * c;
    const size_t tot = (size_t)ne * c;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < tot; t += (size_t)gridDim.x * blockDim.x) {
        const size_t e = CHANNEL_MAJOR ? t % (size_t)ne : t / c;
        const int l = (int)(CHANNEL_MAJOR ? t / (size_t)ne : t % c);
        const int row = idx[e];
        if ((unsigned)row < (unsigned)n) atomicAdd(&dst[CHANNEL_MAJOR ? (size_t)l * n + row : (size_t)row * c + l], src[t]);
    }
}

__global__ void __launch_bounds__(256) grouping_operation_kernel(int c, int n, int m, int ns,
                                                                 const float *__restrict__ feat,
                                                                 const int *__restrict__ idx, float *__restrict__ out)
{
    const int b = blockIdx.y;
    const size_t mk = (size_t)m * ns, tot = (size_t)c * mk;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += (size_t)gridDim.x * blockDim.x) {
        const int ch = (int)(e / mk);
        const size_t jk = e % mk;
        const int ii = idx[(size_t)b * mk + jk];
        out[(size_t)b * tot + e] = feat[((size_t)b * c + ch) * n + ii];
    }
}

static inline unsigned grid_for(size_t tot, int block = 256, unsigned cap = 4096)
{
    size_t g = (tot + block - 1) / block;
    return (unsigned)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace sn

using namespace sn;

// scalar of the sampler step's loss from the per-cloud partials (see sn_sampler_step_loss_forward)
struct StepLossFinal {
    int B, M, N, nproj;
    float w, alpha, lmbda, min_sigma;
    const float *part;
    const float *temperature;
    float *loss;  // NULL: nothing to do
};

// one wave: lane b carries clouds b, b + 64, ...; the lanes are then combined by a fixed xor tree (all loads in flight at
// once; a single thread walking the B partials pays one memory round trip per cloud)
__device__ __forceinline__ void step_loss_final(const StepLossFinal &f, int t)
{
    float s1 = 0.f, mx = 0.f, s2 = 0.f, sp = 0.f;
    for (int b = t; b < f.B; b += 64) s1 += f.part[b * 4], mx += f.part[b * 4 + 1], s2 += f.part[b * 4 + 2], sp += f.part[b * 4 + 3];
    const float T = *f.temperature;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        s1 += __shfl_xor(s1, o);
        mx += __shfl_xor(mx, o);
        s2 += __shfl_xor(s2, o);
        sp += __shfl_xor(sp, o);
    }
    if (t != 0) return;
    const float c12 = s1 / ((float)f.B * (float)f.M), cmax = mx / (float)f.B, c21 = s2 / ((float)f.B * (float)f.N);
    const float lsimp = c12 + cmax + f.w * c21;
    f.loss[0] = f.alpha * lsimp + f.lmbda * sn_sigma(T, f.min_sigma) + sp / ((float)f.B * (float)f.nproj);
    f.loss[1] = lsimp;
}

// Chamfer backward (implicit upstream gradients, targets = the simplified cloud) + soft-projection backward of the same
// query in ONE launch: a wave finishes the Chamfer gradient of target j exactly as chamfer_bwd_reg_kernel does, then runs
// the soft-projection backward of query j (same point: the simplified cloud is both) and stores the sum -- the same
// numbers as the two launches with accumulate_q (chamfer term first, soft term added), one kernel boundary less.
// Optional (engine path: no launch between the pair scan and the backward), keys mode = sn_pairscan_forward_keys +
// sn_sampler_step_loss_keys: the per-point minima are already complete, as INVERTED (distance, query) keys [B][ns]; the argmax
// of dist_q comes from the scan's per-workgroup maxima qmax [B][G]; the y == 0 workgroup leaves sum dist_p of its cloud in
// dpsum [B].  keys == NULL: idxS / argmax come from memory.
struct StepLossFold {
    int G;
    const sn_u64 *keys;
    const sn_u64 *qmax;
    float *dpsum;
};


// Debug build only (-DSN_CS_TIMELINE=1|2, tools/pairscan_timeline.py): thread 0 of every workgroup of chamfer_soft_bwd_kernel
// stamps the 100 MHz wall clock at the phase boundaries of its first query (2: every stamp first drains the memory counters).
#ifdef SN_CS_TIMELINE
__device__ unsigned long long g_cs_tl[8192 * 16];
#define CS_TL(slot)                                                                                              \
    do {                                                                                                         \
        if (threadIdx.x == 0) g_cs_tl[((blockIdx.y * gridDim.x + blockIdx.x) & 8191) * 16 + (slot)] = wall_clock64(); \
    } while (0)
#if SN_CS_TIMELINE >= 2
#define CS_TL_DRAIN() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory")
#else
#define CS_TL_DRAIN()
#endif
#else
#define CS_TL(slot)
#define CS_TL_DRAIN()
#endif

template <int PPL>
__global__ void __launch_bounds__(256) chamfer_soft_bwd_kernel(int nt, int ns, const float *__restrict__ T,
                                                               const float *__restrict__ S, const int *__restrict__ idxT,
                                                               const int *__restrict__ idxS, float *__restrict__ gradT,
                                                               ImplicitGrad ig, SoftBwdArgs sa, StepLossFinal fin,
                                                               StepLossFold fold)
{
    __shared__ float s_part[4];
    __shared__ int s_ip[PPL * 64];
    __shared__ float s_red[3][4];
    __shared__ int s_am;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwaves = blockDim.x >> 6;
    const int b = blockIdx.x;
    CS_TL(0);
    kernarg_warm_for<24, int, int, const float *, const float *, const int *, const int *, float *, ImplicitGrad, SoftBwdArgs,
                     StepLossFinal, StepLossFold>();  // (-0.3 us: see sn_common.h)
    const int nsplit = fin.loss ? (int)gridDim.y - 1 : (int)gridDim.y;  // the last y-slice only combines the loss value
    if ((int)blockIdx.y == nsplit) {
        if (b == 0 && wave == 0) step_loss_final(fin, lane);
        return;
    }
    // ---- everything that does not wait for the keys is requested first (one memory round trip instead of a chain of them:
    // cloud, scalars, the first query's coordinates / nearest point / neighbour indices), the loads that depend on those
    // (the nearest point's and the neighbours' coordinates) while the keys are on their way
    T += (size_t)b * nt * 3, S += (size_t)b * ns * 3;
    idxT += (size_t)b * nt;
    if (idxS) idxS += (size_t)b * ns;
    gradT += (size_t)b * nt * 3;
    float sx[PPL], sy[PPL], sz[PPL], gg[PPL];
    int is[PPL];
#pragma unroll
    for (int i = 0; i < PPL; ++i) {
        const int l = i * 64 + lane;
        const int lc = l < ns ? l : 0;
        const sn_xyz3 sv = *reinterpret_cast<const sn_xyz3 *>(S + (size_t)lc * 3);
        sx[i] = sv.x, sy[i] = sv.y, sz[i] = sv.z;
    }
    const float gLv = *ig.gL * ig.gscale;
    const float Tm = *sa.temperature;
    const float sigma = sn_sigma(Tm, sa.min_sigma);
    const int jfirst = blockIdx.y * nwaves + wave, jstep = nsplit * nwaves;
    // the first query of this wave (the simplified cloud is target and query at once: T = sa.Q of this cloud)
    float tx = 0.f, ty = 0.f, tz = 0.f, ox = 0.f, oy = 0.f, oz = 0.f;
    int j2 = 0;
    SoftQuery sq{};
    auto load_query = [&](int j) {
        tx = T[j], ty = T[j + nt], tz = T[j + 2 * nt];  // targets channel-major (3, nt)
        j2 = idxT[j];
        soft_bwd_load_index(sa, b, j, lane, T, sq);
    };
    auto load_dependent = [&]() {
        ox = S[j2 * 3 + 0], oy = S[j2 * 3 + 1], oz = S[j2 * 3 + 2];
        soft_bwd_load_points(sa, lane, S, sq);
    };
    if (jfirst < nt) load_query(jfirst);
    CS_TL_DRAIN();
    CS_TL(8);

    if (fold.keys) {
        // nearest query of every point of the cloud, fetched ONCE per workgroup (the four waves all need all of them; keys that
        // were updated by device-scope atomics are slow to read: 4 x 8 KB per workgroup cost +4 us) and handed over in LDS
        sn_u64 kv[PPL * 64 / 256 > 0 ? PPL * 64 / 256 : 1];
#pragma unroll
        for (int r = 0; r < (PPL * 64 + 255) / 256; ++r) {
            const int n = threadIdx.x + r * 256;
            // (unconditional: the loads leave as ONE batch -- behind a select each waited for the one before it, 1.16 us
            //  until the keys had landed against 0.36 us now)
            kv[r] = fold.keys[(size_t)b * ns + (n < ns ? n : ns - 1)];
        }
        sn_u64 mk = 0;
        if (wave == 0) {  // argmax of dist_q: maximum of the (dist_q, ~query) keys of the cloud's scan workgroups
            for (int g = lane; g < fold.G; g += 64) {
                const sn_u64 v = fold.qmax[(size_t)b * fold.G + g];
                mk = v > mk ? v : mk;
            }
        }
#if defined(SN_CS_TIMELINE) && SN_CS_TIMELINE >= 2  // (the keys alone)
        CS_TL_DRAIN();
        CS_TL(11);
#endif
        if (jfirst < nt) load_dependent();
        CS_TL_DRAIN();
        CS_TL(9);
        float sdp = 0.f;
#pragma unroll
        for (int r = 0; r < (PPL * 64 + 255) / 256; ++r) {  // (ascending n per thread, as a strided loop would visit them)
            const int n = threadIdx.x + r * 256;
            if (n < ns) {
                const sn_u64 k = ~kv[r];
                s_ip[n] = key_index(k);
                sdp += key_dist(k);
            }
        }
        if (blockIdx.y == 0) {  // sum dist_p of the cloud (for the loss value), fixed order
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) sdp += __shfl_xor(sdp, o);
            if (lane == 0) s_red[2][wave] = sdp;
        }
        if (wave == 0) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const unsigned hi = __shfl_xor((unsigned)(mk >> 32), o), lo = __shfl_xor((unsigned)mk, o);
                const sn_u64 v = ((sn_u64)hi << 32) | lo;
                mk = v > mk ? v : mk;
            }
            if (lane == 0) s_am = (int)(0xFFFFFFFFu - (unsigned)mk);
        }
        CS_TL(10);
        __syncthreads();
        if (blockIdx.y == 0 && threadIdx.x == 0) fold.dpsum[b] = (s_red[2][0] + s_red[2][1]) + (s_red[2][2] + s_red[2][3]);
    } else if (jfirst < nt) {
        load_dependent();
    }
    CS_TL(1);
    const int amt = fold.keys ? s_am : (ig.argmax_t ? ig.argmax_t[b] : -1), ams = ig.argmax_s ? ig.argmax_s[b] : -1;
    float gsig = 0.f;
#pragma unroll
    for (int i = 0; i < PPL; ++i) {
        const int l = i * 64 + lane;
        const int lc = l < ns ? l : 0;
        is[i] = l < ns ? (fold.keys ? s_ip[lc] : idxS[lc]) : -1;
        gg[i] = gLv * (ig.cs + (l == ams ? ig.cmax_s : 0.f)) * 2;
    }
    CS_TL_DRAIN();
    CS_TL(2);

    for (int j = jfirst; j < nt; j += jstep) {
        if (j != jfirst) {
            load_query(j);
            load_dependent();
        }
        const float g = gLv * (ig.ct + (j == amt ? ig.cmax_t : 0.f)) * 2;
        float ax = g * (tx - ox), ay = g * (ty - oy), az = g * (tz - oz);  // own term first
        if (j == (int)blockIdx.y * nwaves) { CS_TL_DRAIN(); CS_TL(3); }
#pragma unroll
        for (int i = 0; i < PPL; ++i) {
            sn_u64 mask = __ballot(is[i] == j);
            if (mask) {
                const float cx = gg[i] * (sx[i] - tx), cy = gg[i] * (sy[i] - ty), cz = gg[i] * (sz[i] - tz);
                while (mask) {
                    const int t = __builtin_ctzll(mask);
                    mask &= mask - 1;
                    ax -= readlane_f(cx, t);
                    ay -= readlane_f(cy, t);
                    az -= readlane_f(cz, t);
                }
            }
        }
        if (j == (int)blockIdx.y * nwaves) CS_TL(4);
        float qx, qy, qz, asg;
        soft_bwd_math<true>(sa, b, j, lane, sigma, sq, qx, qy, qz, asg);
        if (j == (int)blockIdx.y * nwaves) CS_TL(5);
        gsig += asg;
        if (lane < 3) gradT[j + lane * nt] = lane == 0 ? ax + qx : (lane == 1 ? ay + qy : az + qz);
    }
    if (lane == 0) s_part[wave] = gsig;
    __syncthreads();
    if (threadIdx.x == 0) {
        float tot = 0.f;
        for (int w2 = 0; w2 < nwaves; ++w2) tot += s_part[w2];
        sa.grad_sigma_partial[(size_t)b * nsplit + blockIdx.y] = tot;
    }
    CS_TL_DRAIN();
    CS_TL(6);
}

// workgroups per cloud of the soft-projection backward kernels; grad_sigma_partial holds b * this many floats
extern "C" int sn_soft_bwd_splits(int b, int m)
{
    // query slices per cloud of the soft-projection / fused loss backward: ~1024 workgroups in all (swept 512 / 1024 / 2048 / 4096
    // at 64 .. 2048 clouds: 1024 is fastest everywhere -- chamfer_soft_bwd_kernel 57.5 -> 43.8 us at 512 clouds against 512; a wave
    // walks its queries one after the other, each behind two dependent memory round trips)
    return std::max(1, std::min((m + 3) / 4, (kChamferBwdGroups + b - 1) / std::max(b, 1)));
}

extern "C" int sn_chamfer_backward(int b, int n, const float *xyz1, int m, const float *xyz2,
                                   const float *grad_dist1, const int *idx1, const float *grad_dist2,
                                   const int *idx2, float *grad_xyz1, float *grad_xyz2, sn_stream_t stream)
{
    SN_REQUIRE(b >= 0 && n >= 0 && m >= 0, "negative size");
    if (b == 0 || n == 0 || m == 0) return 0;
    SN_REQUIRE(xyz1 && xyz2 && grad_dist1 && grad_dist2 && idx1 && idx2, "null input");
    hipStream_t st = (hipStream_t)stream;
    auto ysplit = [&](int nt) { return std::max(1, std::min((nt + 3) / 4, (kChamferBwdGroups + b - 1) / b)); };
    const ImplicitGrad none{};
    if (grad_xyz1) launch_chamfer_bwd(b, ysplit(n), n, m, xyz1, xyz2, grad_dist1, idx1, grad_dist2, idx2, grad_xyz1, 1, none, st);
    if (grad_xyz2) launch_chamfer_bwd(b, ysplit(m), m, n, xyz2, xyz1, grad_dist2, idx2, grad_dist1, idx1, grad_xyz2, 0, none, st);
    SN_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// Fused simplification loss (samplenet.py:171-181):
//   loss = mean(dist1) + mean_b(max_m dist1) + w * mean(dist2),   w = gamma + delta * pc_size
// forward: one workgroup per cloud reduces its (sum dist1, max dist1 + first argmax, sum dist2) in a fixed order,
// a single-workgroup kernel combines the clouds; backward: sn_chamfer_backward with implicit upstream gradients.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) simp_loss_partial_kernel(int n1, int n2, const float *__restrict__ d1,
                                                                const float *__restrict__ d2, float *__restrict__ part,
                                                                int *__restrict__ argmax1)
{
    __shared__ float r0[256], r1[256], r2[256];
    __shared__ int ri[256];
    const int b = blockIdx.x, t = threadIdx.x;
    float s1 = 0.f, mx = -INFINITY, s2 = 0.f;
    int am = 0;
    for (int j = t; j < n1; j += 256) {
        const float v = d1[(size_t)b * n1 + j];
        s1 += v;
        if (v > mx) mx = v, am = j;
    }
    for (int j = t; j < n2; j += 256) s2 += d2[(size_t)b * n2 + j];
    r0[t] = s1, r1[t] = mx, r2[t] = s2, ri[t] = am;
    for (int s = 128; s > 0; s >>= 1) {
        __syncthreads();
        if (t < s) {
            r0[t] += r0[t + s];
            r2[t] += r2[t + s];
            if (r1[t + s] > r1[t] || (r1[t + s] == r1[t] && ri[t + s] < ri[t])) r1[t] = r1[t + s], ri[t] = ri[t + s];
        }
    }
    if (t == 0) {
        part[b * 3 + 0] = r0[0], part[b * 3 + 1] = r1[0], part[b * 3 + 2] = r2[0];
        argmax1[b] = ri[0];
    }
}

template <bool WITH_MAX>
__global__ void __launch_bounds__(64) simp_loss_final_kernel(int B, int n1, int n2, float w, const float *__restrict__ part,
                                                             float *__restrict__ loss)
{
    if (threadIdx.x != 0) return;
    float s1 = 0.f, mx = 0.f, s2 = 0.f;
    for (int b = 0; b < B; ++b) s1 += part[b * 3], mx += part[b * 3 + 1], s2 += part[b * 3 + 2];
    const float c12 = s1 / ((float)B * (float)n1), cmax = mx / (float)B, c21 = s2 / ((float)B * (float)n2);
    loss[0] = WITH_MAX ? c12 + cmax + w * c21 : c12 + w * c21;
}

extern "C" int sn_simplification_loss_forward(int B, int n1, int n2, const float *dist1, const float *dist2, float weight,
                                              float *partial, int *argmax1, float *loss, sn_stream_t stream)
{
    SN_REQUIRE(B >= 1 && n1 >= 1 && n2 >= 1, "bad size");
    SN_REQUIRE(dist1 && dist2 && partial && argmax1 && loss, "null pointer");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(simp_loss_partial_kernel, dim3(B), dim3(256), 0, st, n1, n2, dist1, dist2, partial, argmax1);
    hipLaunchKernelGGL(simp_loss_final_kernel<true>, dim3(1), dim3(64), 0, st, B, n1, n2, weight, partial, loss);
    SN_LAUNCH_CHECK();
    return 0;
}

// grad_xyz1 (B,n1,3) / grad_xyz2 (B,n2,3) of the fused loss; grad_loss: device scalar.  Either output may be NULL.
extern "C" int sn_simplification_loss_backward(int B, int n1, const float *xyz1, int n2, const float *xyz2, const int *idx1,
                                               const int *idx2, const int *argmax1, float weight, const float *grad_loss,
                                               float *grad_xyz1, float *grad_xyz2, int layout1, sn_stream_t stream)
{
    SN_REQUIRE(layout1 == 0 || layout1 == 1, "layout1 must be 0 (B,n1,3) or 1 (B,3,n1)");
    SN_REQUIRE(layout1 == 0 || !grad_xyz2, "grad_xyz2 needs xyz1 in (B,n1,3) layout");
    SN_REQUIRE(B >= 1 && n1 >= 1 && n2 >= 1, "bad size");
    SN_REQUIRE(xyz1 && xyz2 && idx1 && idx2 && argmax1 && grad_loss, "null pointer");
    hipStream_t st = (hipStream_t)stream;
    auto ysplit = [&](int nt) { return std::max(1, std::min((nt + 3) / 4, (kChamferBwdGroups + B - 1) / B)); };
    const float c1 = 1.0f / ((float)B * (float)n1), cm = 1.0f / (float)B, c2 = weight / ((float)B * (float)n2);
    if (grad_xyz1) {
        ImplicitGrad ig{grad_loss, argmax1, nullptr, c1, cm, c2, 0.f};
        launch_chamfer_bwd(B, ysplit(n1), n1, n2, xyz1, xyz2, nullptr, idx1, nullptr, idx2, grad_xyz1, 1, ig, st, layout1);
    }
    if (grad_xyz2) {
        ImplicitGrad ig{grad_loss, nullptr, argmax1, c2, 0.f, c1, cm};
        launch_chamfer_bwd(B, ysplit(n2), n2, n1, xyz2, xyz1, nullptr, idx2, nullptr, idx1, grad_xyz2, 0, ig, st);
    }
    SN_LAUNCH_CHECK();
    return 0;
}

// The Chamfer loss of the registration task (registration/main.py:573-577): mean(dist1) + mean(dist2) -- the simplification loss
// without its maximum term; same kernels (partial: 3 B floats, argmax1: B ints of scratch), same implicit-gradient backward.
extern "C" int sn_chamfer_mean_loss_forward(int B, int n1, int n2, const float *dist1, const float *dist2, float *partial,
                                            int *argmax1, float *loss, sn_stream_t stream)
{
    SN_REQUIRE(B >= 1 && n1 >= 1 && n2 >= 1, "bad size");
    SN_REQUIRE(dist1 && dist2 && partial && argmax1 && loss, "null pointer");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(simp_loss_partial_kernel, dim3(B), dim3(256), 0, st, n1, n2, dist1, dist2, partial, argmax1);
    hipLaunchKernelGGL(simp_loss_final_kernel<false>, dim3(1), dim3(64), 0, st, B, n1, n2, 1.0f, partial, loss);
    SN_LAUNCH_CHECK();
    return 0;
}

extern "C" int sn_chamfer_mean_loss_backward(int B, int n1, const float *xyz1, int n2, const float *xyz2, const int *idx1,
                                             const int *idx2, const float *grad_loss, float *grad_xyz1, float *grad_xyz2,
                                             sn_stream_t stream)
{
    SN_REQUIRE(B >= 1 && n1 >= 1 && n2 >= 1, "bad size");
    SN_REQUIRE(xyz1 && xyz2 && idx1 && idx2 && grad_loss, "null pointer");
    hipStream_t st = (hipStream_t)stream;
    auto ysplit = [&](int nt) { return std::max(1, std::min((nt + 3) / 4, (kChamferBwdGroups + B - 1) / B)); };
    const float c1 = 1.0f / ((float)B * (float)n1), c2 = 1.0f / ((float)B * (float)n2);
    if (grad_xyz1) {
        ImplicitGrad ig{grad_loss, nullptr, nullptr, c1, 0.f, c2, 0.f};
        launch_chamfer_bwd(B, ysplit(n1), n1, n2, xyz1, xyz2, nullptr, idx1, nullptr, idx2, grad_xyz1, 1, ig, st);
    }
    if (grad_xyz2) {
        ImplicitGrad ig{grad_loss, nullptr, nullptr, c2, 0.f, c1, 0.f};
        launch_chamfer_bwd(B, ysplit(n2), n2, n1, xyz2, xyz1, nullptr, idx2, nullptr, idx1, grad_xyz2, 0, ig, st);
    }
    SN_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// The Chamfer term of the registration task loss for SEVERAL evaluations in one batch (the progressive sampler's prefixes against one
// template: clouds [e group, (e + 1) group) are evaluation e): xyz1 (R, n1, 3) holds clouds of nv[e] <= n1 valid points each -- the rest
// are cyclic copies (sn_cyclic_pad_cat), which the scan may see (a copy never wins a minimum: the lowest index does) but the loss must
// not count --, xyz2 (R, n2, 3).  loss[e] = mean over the evaluation's clouds and VALID points of dist1 + mean of dist2, reduced in
// sn_chamfer_mean_loss_forward's order; the backward is chamfer_bwd_reg_kernel's with the evaluation's own upstream gradient and
// 1 / (group nv[e]), copies as targets get zero and as sources are skipped: every number equals the evaluation's own
// sn_chamfer_mean_loss_forward / _backward on the unpadded cloud, bit for bit -- 2 + 2 launches where E evaluations take 2 E + 2 E.
// ------------------------------------------------------------------------------------------------
struct GroupSizes {
    int n;
    int nv[kMaxPrefixes];
};
__global__ void __launch_bounds__(256) chamfer_mean_grouped_partial_kernel(int n1, int n2, int group, GroupSizes gs,
                                                                           const float *__restrict__ d1, const float *__restrict__ d2,
                                                                           float *__restrict__ part)
{
    __shared__ float r0[256], r2[256];
    const int b = blockIdx.x, t = threadIdx.x;
    const int nv = gs.nv[b / group];
    float s1 = 0.f, s2 = 0.f;
    for (int j = t; j < nv; j += 256) s1 += d1[(size_t)b * n1 + j];
    for (int j = t; j < n2; j += 256) s2 += d2[(size_t)b * n2 + j];
    r0[t] = s1, r2[t] = s2;
    for (int s = 128; s > 0; s >>= 1) {
        __syncthreads();
        if (t < s) r0[t] += r0[t + s], r2[t] += r2[t + s];
    }
    if (t == 0) part[b * 2] = r0[0], part[b * 2 + 1] = r2[0];
}
__global__ void __launch_bounds__(64) chamfer_mean_grouped_final_kernel(int n2, int group, GroupSizes gs, const float *__restrict__ part,
                                                                        float *__restrict__ loss)
{
    const int e = threadIdx.x;
    if (e >= gs.n) return;
    float s1 = 0.f, s2 = 0.f;
    for (int b = e * group; b < (e + 1) * group; ++b) s1 += part[b * 2], s2 += part[b * 2 + 1];
    const float c12 = s1 / ((float)group * (float)gs.nv[e]), c21 = s2 / ((float)group * (float)n2);
    loss[e] = c12 + 1.0f * c21;
}

static int group_sizes(int R, int n1, int group, int nev, const int *nvalid, GroupSizes &gs)
{
    SN_REQUIRE(group >= 1 && nev >= 1 && nev <= kMaxPrefixes && R == nev * group && nvalid, "R = nev * group, at most 16 evaluations");
    gs.n = nev;
    for (int e = 0; e < nev; ++e) {
        SN_REQUIRE(nvalid[e] >= 1 && nvalid[e] <= n1, "valid points outside [1, n1]");
        gs.nv[e] = nvalid[e];
    }
    return 0;
}

// dist1 (R, n1), dist2 (R, n2): sn_chamfer_forward's products of the padded batch; partial: 2 R floats; loss: nev floats
extern "C" int sn_chamfer_mean_loss_forward_grouped(int R, int n1, int n2, int group, int nev, const int *nvalid, const float *dist1,
                                                    const float *dist2, float *partial, float *loss, sn_stream_t stream)
{
    SN_REQUIRE(R >= 1 && n1 >= 1 && n2 >= 1 && dist1 && dist2 && partial && loss, "bad argument");
    GroupSizes gs{};
    if (int rc = group_sizes(R, n1, group, nev, nvalid, gs)) return rc;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(chamfer_mean_grouped_partial_kernel, dim3(R), dim3(256), 0, st, n1, n2, group, gs, dist1, dist2, partial);
    hipLaunchKernelGGL(chamfer_mean_grouped_final_kernel, dim3(1), dim3(64), 0, st, n2, group, gs, partial, loss);
    SN_LAUNCH_CHECK();
    return 0;
}

// chamfer_bwd_reg_kernel for grouped evaluations: gL one upstream gradient per evaluation; ntv / nsv = the evaluation's valid targets /
// sources (the other side: all of them); ct / cs as sn_chamfer_mean_loss_backward forms them, per evaluation
struct GroupedGrad {
    const float *gL;
    int group;
    int t_valid;  // 1: the TARGET side is the padded one (nv[] bounds the targets), 0: the source side
    GroupSizes gs;
    float c_pad[kMaxPrefixes];  // 1 / (group nv[e]): the coefficient of the padded side
    float c_full;               // 1 / (group n_other)
};
template <int PPL>
__global__ void __launch_bounds__(256) chamfer_bwd_grouped_kernel(int nt, int ns, const float *__restrict__ T, const float *__restrict__ S,
                                                                  const int *__restrict__ idxT, const int *__restrict__ idxS,
                                                                  float *__restrict__ gradT, int own_first, GroupedGrad gg)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwaves = blockDim.x >> 6;
    const int b = blockIdx.x, e = b / gg.group;
    T += (size_t)b * nt * 3, S += (size_t)b * ns * 3;
    idxT += (size_t)b * nt, idxS += (size_t)b * ns;
    gradT += (size_t)b * nt * 3;
    const float gLv = gg.gL[e] * 1.f;
    const float ct = gg.t_valid ? gg.c_pad[e] : gg.c_full, cs = gg.t_valid ? gg.c_full : gg.c_pad[e];
    const int ntv = gg.t_valid ? gg.gs.nv[e] : nt, nsv = gg.t_valid ? ns : gg.gs.nv[e];
    float sx[PPL], sy[PPL], sz[PPL], gs2[PPL];
    int is[PPL];
#pragma unroll
    for (int i = 0; i < PPL; ++i) {
        const int l = i * 64 + lane;
        const int lc = l < ns ? l : 0;
        const sn_xyz3 sv = *reinterpret_cast<const sn_xyz3 *>(S + (size_t)lc * 3);
        sx[i] = sv.x, sy[i] = sv.y, sz[i] = sv.z;
        is[i] = l < nsv ? idxS[lc] : -1;  // (copies among the sources are not counted)
        gs2[i] = (gLv * (cs + 0.f)) * 2;
    }
    for (int j = blockIdx.y * nwaves + wave; j < nt; j += gridDim.y * nwaves) {
        if (j >= ntv) {  // a copy among the targets: no gradient
            if (lane < 3) gradT[j * 3 + lane] = 0.f;
            continue;
        }
        const float tx = T[j * 3], ty = T[j * 3 + 1], tz = T[j * 3 + 2];
        const int j2 = idxT[j];
        const float g = (gLv * (ct + 0.f)) * 2;
        const float ox = g * (tx - S[j2 * 3 + 0]);
        const float oy = g * (ty - S[j2 * 3 + 1]);
        const float oz = g * (tz - S[j2 * 3 + 2]);
        float ax = 0.f, ay = 0.f, az = 0.f;
        if (own_first) ax += ox, ay += oy, az += oz;
#pragma unroll
        for (int i = 0; i < PPL; ++i) {
            sn_u64 mask = __ballot(is[i] == j);
            if (mask) {
                const float cx = gs2[i] * (sx[i] - tx), cy = gs2[i] * (sy[i] - ty), cz = gs2[i] * (sz[i] - tz);
                while (mask) {
                    const int t = __builtin_ctzll(mask);
                    mask &= mask - 1;
                    ax -= readlane_f(cx, t);
                    ay -= readlane_f(cy, t);
                    az -= readlane_f(cz, t);
                }
            }
        }
        if (!own_first) ax += ox, ay += oy, az += oz;
        if (lane < 3) gradT[j * 3 + lane] = lane == 0 ? ax : (lane == 1 ? ay : az);
    }
}

// grad_xyz1 (R, n1, 3) [zero at the copies] and / or grad_xyz2 (R, n2, 3); grad_loss: nev floats on the device; n1, n2 <= 2048
extern "C" int sn_chamfer_mean_loss_backward_grouped(int R, int n1, const float *xyz1, int n2, const float *xyz2, int group, int nev,
                                                     const int *nvalid, const int *idx1, const int *idx2, const float *grad_loss,
                                                     float *grad_xyz1, float *grad_xyz2, sn_stream_t stream)
{
    SN_REQUIRE(R >= 1 && n1 >= 1 && n2 >= 1 && n1 <= 2048 && n2 <= 2048, "bad size (at most 2048 points a side)");
    SN_REQUIRE(xyz1 && xyz2 && idx1 && idx2 && grad_loss, "null pointer");
    GroupedGrad gg{};
    if (int rc = group_sizes(R, n1, group, nev, nvalid, gg.gs)) return rc;
    gg.gL = grad_loss, gg.group = group;
    for (int e = 0; e < nev; ++e) gg.c_pad[e] = 1.0f / ((float)group * (float)nvalid[e]);
    gg.c_full = 1.0f / ((float)group * (float)n2);
    hipStream_t st = (hipStream_t)stream;
    auto ysplit = [&](int nt) { return std::max(1, std::min((nt + 3) / 4, (kChamferBwdGroups + R - 1) / R)); };
#define SN_CBG(PPL_, NT, NS, TT, SS, IT, IS, GT, OF)                                                                              \
    hipLaunchKernelGGL(chamfer_bwd_grouped_kernel<PPL_>, dim3(R, ysplit(NT)), dim3(256), 0, st, NT, NS, TT, SS, IT, IS, GT, OF, gg)
#define SN_CBG_ANY(NT, NS, TT, SS, IT, IS, GT, OF)           \
    do {                                                     \
        if (NS <= 64) SN_CBG(1, NT, NS, TT, SS, IT, IS, GT, OF);        \
        else if (NS <= 256) SN_CBG(4, NT, NS, TT, SS, IT, IS, GT, OF);  \
        else if (NS <= 1024) SN_CBG(16, NT, NS, TT, SS, IT, IS, GT, OF); \
        else SN_CBG(32, NT, NS, TT, SS, IT, IS, GT, OF);                \
    } while (0)
    if (grad_xyz1) {
        gg.t_valid = 1;
        SN_CBG_ANY(n1, n2, xyz1, xyz2, idx1, idx2, grad_xyz1, 1);
    }
    if (grad_xyz2) {
        gg.t_valid = 0;
        SN_CBG_ANY(n2, n1, xyz2, xyz1, idx2, idx1, grad_xyz2, 0);
    }
#undef SN_CBG_ANY
#undef SN_CBG
    SN_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// PCRNet's output head (registration/models/pcrnet.py:78-82 + the QuaterNet regulariser of registration/main.py:565):
//   twist (B,7) = [ y[:, 0:4] / max(||y[:, 0:4]||, 1e-12) | y[:, 4:7] ],   qnorm = mean_b (||y[:, 0:4]||^2 - 1)^2
// one single-workgroup launch forward, one backward (torch: slice, norm, clamp, expand, div, cat + pow, sum, sub, pow, mean and
// their ~15 backward launches).  Sums over the batch run in a fixed order.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) pcrnet_head_fwd_kernel(int B, const float *__restrict__ y, float *__restrict__ twist,
                                                              float *__restrict__ quat, float *__restrict__ qnorm)
{
    __shared__ float red[256];
    float acc = 0.f;
    for (int b = threadIdx.x; b < B; b += 256) {
        const float *r = y + (size_t)b * 7;
        const float n2 = r[0] * r[0] + r[1] * r[1] + r[2] * r[2] + r[3] * r[3];
        const float inv = 1.0f / fmaxf(sqrtf(n2), 1e-12f);
        const float q0 = r[0] * inv, q1 = r[1] * inv, q2 = r[2] * inv, q3 = r[3] * inv;
        float *o = twist + (size_t)b * 7;
        o[0] = q0, o[1] = q1, o[2] = q2, o[3] = q3, o[4] = r[4], o[5] = r[5], o[6] = r[6];
        if (quat) quat[b * 4 + 0] = q0, quat[b * 4 + 1] = q1, quat[b * 4 + 2] = q2, quat[b * 4 + 3] = q3;
        acc += (n2 - 1.0f) * (n2 - 1.0f);
    }
    if (!qnorm) return;
    red[threadIdx.x] = acc;
    for (int s = 128; s > 0; s >>= 1) {
        __syncthreads();
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    }
    if (threadIdx.x == 0) qnorm[0] = red[0] / (float)B;
}

__global__ void __launch_bounds__(256) pcrnet_head_bwd_kernel(int B, const float *__restrict__ y, const float *__restrict__ g_twist,
                                                              const float *__restrict__ g_quat, const float *__restrict__ g_qnorm,
                                                              float *__restrict__ g_y)
{
    const float gq = g_qnorm ? g_qnorm[0] : 0.f;
    for (int b = threadIdx.x; b < B; b += 256) {
        const float *r = y + (size_t)b * 7;
        const float n2 = r[0] * r[0] + r[1] * r[1] + r[2] * r[2] + r[3] * r[3];
        const float nrm = sqrtf(n2);
        float *o = g_y + (size_t)b * 7;
        // upstream gradient of the normalised quaternion: through twist[:, 0:4] and / or through the separate copy
        float gt[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 4; ++i) gt[i] = (g_twist ? g_twist[(size_t)b * 7 + i] : 0.f) + (g_quat ? g_quat[b * 4 + i] : 0.f);
        float gp[4];
        if (nrm > 1e-12f) {  // d(p / ||p||) = (g - q (q . g)) / ||p||
            const float inv = 1.0f / nrm;
            const float q[4] = {r[0] * inv, r[1] * inv, r[2] * inv, r[3] * inv};
            const float dot = q[0] * gt[0] + q[1] * gt[1] + q[2] * gt[2] + q[3] * gt[3];
#pragma unroll
            for (int i = 0; i < 4; ++i) gp[i] = (gt[i] - q[i] * dot) * inv;
        } else {  // clamped denominator: p / 1e-12
#pragma unroll
            for (int i = 0; i < 4; ++i) gp[i] = gt[i] * 1e12f;
        }
#pragma unroll
        for (int i = 4; i < 7; ++i) o[i] = g_twist ? g_twist[(size_t)b * 7 + i] : 0.f;
        const float cq = gq * 4.0f * (n2 - 1.0f) / (float)B;  // d/dp mean (||p||^2 - 1)^2 = 4 (||p||^2 - 1) p / B
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = gp[i] + cq * r[i];
    }
}

extern "C" int sn_pcrnet_head_forward(int B, const float *y, float *twist, float *quat, float *qnorm, sn_stream_t stream)
{
    SN_REQUIRE(B >= 1 && y && twist, "bad argument");
    hipLaunchKernelGGL(pcrnet_head_fwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, B, y, twist, quat, qnorm);
    SN_LAUNCH_CHECK();
    return 0;
}

extern "C" int sn_pcrnet_head_backward(int B, const float *y, const float *g_twist, const float *g_quat, const float *g_qnorm, float *g_y,
                                       sn_stream_t stream)
{
    SN_REQUIRE(B >= 1 && y && g_y, "bad argument");
    hipLaunchKernelGGL(pcrnet_head_bwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, B, y, g_twist, g_quat, g_qnorm, g_y);
    SN_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// The sampler's total loss as registration/main.py:507-531 composes it, with the benchmark's stand-in task term
// (SURVEY.md 8d):   L = alpha * L_simp + lmbda * max(T^2, min_sigma) + mean(proj)
// One single-workgroup kernel forward, one backward (replaces ~20 elementwise / reduction launches of the op-by-op
// composition).  Reductions run in a fixed order.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024) sampler_loss_fwd_kernel(int nproj, const float *__restrict__ proj,
                                                                const float *__restrict__ lsimp,
                                                                const float *__restrict__ temperature, float alpha,
                                                                float lmbda, float min_sigma, float *__restrict__ loss)
{
    __shared__ float red[1024];
    float acc = 0.f;
    for (int i = threadIdx.x; i < nproj; i += 1024) acc += proj[i];
    red[threadIdx.x] = acc;
    for (int s = 512; s > 0; s >>= 1) {
        __syncthreads();
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    }
    if (threadIdx.x == 0) {
        const float T = *temperature;
        loss[0] = alpha * lsimp[0] + lmbda * sn_sigma(T, min_sigma) + red[0] / (float)nproj;
    }
}

__global__ void __launch_bounds__(256) sampler_loss_bwd_kernel(int nproj, const float *__restrict__ grad_loss,
                                                               const float *__restrict__ temperature, float alpha,
                                                               float lmbda, float min_sigma, float *__restrict__ grad_proj,
                                                               float *__restrict__ grad_lsimp, float *__restrict__ grad_T)
{
    const float g = grad_loss[0];
    const float gp = g / (float)nproj;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nproj; i += gridDim.x * blockDim.x) grad_proj[i] = gp;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        grad_lsimp[0] = alpha * g;
        const float T = *temperature, t2 = T * T;
        const float w = t2 > min_sigma ? 1.f : (t2 == min_sigma ? 0.5f : 0.f);
        grad_T[0] = lmbda * g * w * 2.0f * T;
    }
}

extern "C" int sn_sampler_loss_forward(int nproj, const float *proj, const float *lsimp, const float *temperature,
                                       float alpha, float lmbda, float min_sigma, float *loss, sn_stream_t stream)
{
    SN_REQUIRE(nproj >= 1 && proj && lsimp && temperature && loss, "bad argument");
    hipLaunchKernelGGL(sampler_loss_fwd_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, nproj, proj, lsimp, temperature,
                       alpha, lmbda, min_sigma, loss);
    SN_LAUNCH_CHECK();
    return 0;
}

extern "C" int sn_sampler_loss_backward(int nproj, const float *grad_loss, const float *temperature, float alpha,
                                        float lmbda, float min_sigma, float *grad_proj, float *grad_lsimp, float *grad_T,
                                        sn_stream_t stream)
{
    SN_REQUIRE(nproj >= 1 && grad_loss && temperature && grad_proj && grad_lsimp && grad_T, "bad argument");
    hipLaunchKernelGGL(sampler_loss_bwd_kernel, dim3((nproj + 255) / 256), dim3(256), 0, (hipStream_t)stream, nproj, grad_loss,
                       temperature, alpha, lmbda, min_sigma, grad_proj, grad_lsimp, grad_T);
    SN_LAUNCH_CHECK();
    return 0;
}

// d loss / dT from the per-workgroup partials of d loss / d sigma:  sigma = max(T^2, min_sigma)
//   dT = (sum partial) * 2T * [T^2 > min_sigma]   (torch.max splits the gradient evenly on an exact tie)
__global__ void __launch_bounds__(256) sigma_grad_kernel(int nparts, const float *__restrict__ partial,
                                                         const float *__restrict__ temperature, float min_sigma,
                                                         float *__restrict__ grad_T, const float *__restrict__ gsigma_direct,
                                                         float direct_scale, StepLossFinal fin = StepLossFinal{},
                                                         sn::StepLossKeysFinal kf = sn::StepLossKeysFinal{})
{
    if (fin.loss && blockIdx.x == 1) {  // second workgroup (fold mode): the step's loss value from the per-cloud partials
        if (threadIdx.x < 64) step_loss_final(fin, threadIdx.x);
        return;
    }
    if (kf.loss && blockIdx.x >= 1) {  // keys mode: workgroup 1 combines the loss value, the others re-zero the key table
        if (blockIdx.x == 1) {
            if (threadIdx.x < 64) sn::step_loss_keys_final(kf, threadIdx.x);
        } else {
            const long long nb = gridDim.x - 2, per = (kf.nkeys + nb - 1) / nb;
            const long long i0 = (long long)(blockIdx.x - 2) * per, i1 = i0 + per < kf.nkeys ? i0 + per : kf.nkeys;
            for (long long i = i0 + threadIdx.x; i < i1; i += 256) kf.keys[i] = 0;
        }
        return;
    }
    __shared__ float red[4];
    sn::sigma_grad_block(nparts, partial, temperature, min_sigma, grad_T, gsigma_direct, direct_scale, red);
}

// sigma = max(T^2, min_sigma) (soft_projection.py:97-99) as a device scalar: the projection loss of a script, one launch
// (torch: pow + maximum forward, seven launches backward; sn_sigma_grad with the upstream gradient as its one partial is the backward)
__global__ void sigma_forward_kernel(const float *__restrict__ temperature, float min_sigma, float *__restrict__ sigma)
{
    if (threadIdx.x == 0) sigma[0] = sn_sigma(temperature[0], min_sigma);
}
extern "C" int sn_sigma_forward(const float *temperature, float min_sigma, float *sigma, sn_stream_t stream)
{
    SN_REQUIRE(temperature && sigma, "null pointer");
    hipLaunchKernelGGL(sigma_forward_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, temperature, min_sigma, sigma);
    SN_LAUNCH_CHECK();
    return 0;
}

extern "C" int sn_sigma_grad(int nparts, const float *partial, const float *temperature, float min_sigma, float *grad_T,
                             sn_stream_t stream)
{
    SN_REQUIRE(nparts >= 1 && partial && temperature && grad_T, "bad argument");
    hipLaunchKernelGGL(sigma_grad_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, nparts, partial, temperature,
                       min_sigma, grad_T, (const float *)nullptr, 0.f);
    SN_LAUNCH_CHECK();
    return 0;
}

extern "C" int sn_soft_project_backward(int b, int n, int m, int k, const float *P, int p_layout, const float *Q,
                                        int q_layout, const int *idx, const float *temperature, float min_sigma,
                                        const float *grad_proj, int gproj_layout, float *grad_Q, int gq_layout,
                                        float *grad_P, float *grad_sigma_partial, sn_stream_t stream)
{
    SN_REQUIRE(b >= 0 && n >= 1 && m >= 0 && k >= 1 && k <= 64, "bad size");
    if (b == 0 || m == 0) return 0;
    SN_REQUIRE(P && Q && idx && temperature && grad_proj, "null input");
    SoftBwdArgs a{};
    a.P = P, a.Q = Q, a.idx = idx, a.temperature = temperature, a.min_sigma = min_sigma;
    a.p_layout = p_layout, a.q_layout = q_layout, a.n = n, a.m = m, a.k = k;
    a.grad_proj = grad_proj, a.gproj_layout = gproj_layout;
    a.grad_Q = grad_Q, a.gq_layout = gq_layout, a.grad_P = grad_P, a.grad_sigma_partial = grad_sigma_partial;
    hipLaunchKernelGGL(soft_bwd_kernel<true>, dim3(b, sn_soft_bwd_splits(b, m)), dim3(256), 0, (hipStream_t)stream, a);
    SN_LAUNCH_CHECK();
    return 0;
}

extern "C" int sn_soft_weights_forward(int b, int n, int m, int k, const float *P, const float *Q, const int *idx,
                                       const float *temperature, float min_sigma, float *weights,
                                       sn_stream_t stream)
{
    SN_REQUIRE(b >= 0 && n >= 1 && m >= 0 && k >= 1 && k <= 64, "bad size");
    if (b == 0 || m == 0) return 0;
    SN_REQUIRE(P && Q && idx && temperature && weights, "null pointer");
    const int y = std::max(1, std::min((m + 3) / 4, (1024 + b - 1) / b));
    hipLaunchKernelGGL(soft_weights_fwd_kernel, dim3(b, y), dim3(256), 0, (hipStream_t)stream, n, m, k, P, Q, idx,
                       temperature, min_sigma, weights);
    SN_LAUNCH_CHECK();
    return 0;
}

extern "C" int sn_soft_weights_backward(int b, int n, int m, int k, const float *P, const float *Q, const int *idx,
                                        const float *temperature, float min_sigma, const float *weights,
                                        const float *grad_weights, float *grad_Q, float *grad_P,
                                        float *grad_sigma_partial, sn_stream_t stream)
{
    SN_REQUIRE(b >= 0 && n >= 1 && m >= 0 && k >= 1 && k <= 64, "bad size");
    if (b == 0 || m == 0) return 0;
    SN_REQUIRE(P && Q && idx && temperature && grad_weights, "null input");
    SoftBwdArgs a{};
    a.P = P, a.Q = Q, a.idx = idx, a.temperature = temperature, a.min_sigma = min_sigma;
    a.p_layout = SN_LAYOUT_BCN, a.q_layout = SN_LAYOUT_BCN, a.n = n, a.m = m, a.k = k;
    a.weights_in = weights, a.grad_weights = grad_weights;
    a.grad_Q = grad_Q, a.gq_layout = SN_LAYOUT_BCN, a.grad_P = grad_P, a.grad_sigma_partial = grad_sigma_partial;
    hipLaunchKernelGGL(soft_bwd_kernel<false>, dim3(b, sn_soft_bwd_splits(b, m)), dim3(256), 0, (hipStream_t)stream, a);
    SN_LAUNCH_CHECK();
    return 0;
}

extern "C" int sn_weighted_gather_forward(int b, int c, int n, int m, int k, const float *X, const int *idx,
                                          const float *weights, float *out, sn_stream_t stream)
{
    SN_REQUIRE(b >= 0 && c >= 0 && n >= 1 && m >= 0 && k >= 1, "bad size");
    if (b == 0 || m == 0 || c == 0) return 0;
    SN_REQUIRE(X && idx && weights && out, "null pointer");
    hipLaunchKernelGGL(weighted_gather_fwd_kernel, dim3(((size_t)c * m + 255) / 256, b), dim3(256), 0,
                       (hipStream_t)stream, c, n, m, k, X, idx, weights, out);
    SN_LAUNCH_CHECK();
    return 0;
}

extern "C" int sn_weighted_gather_backward(int b, int c, int n, int m, int k, const float *X, const int *idx,
                                           const float *weights, const float *grad_out, float *grad_weights,
                                           float *grad_X, sn_stream_t stream)
{
    SN_REQUIRE(b >= 0 && c >= 0 && n >= 1 && m >= 0 && k >= 1, "bad size");
    if (b == 0 || m == 0) return 0;
    SN_REQUIRE(X && idx && weights && grad_out, "null pointer");
    hipLaunchKernelGGL(weighted_gather_bwd_kernel, dim3(((size_t)m * k + 255) / 256, b), dim3(256), 0,
                       (hipStream_t)stream, c, n, m, k, X, idx, weights, grad_out, grad_weights, grad_X);
    SN_LAUNCH_CHECK();
    return 0;
}

// dst (b, n, c) [or (b, c, n)] = index-add of src over idx (b, ne): ordered and deterministic up to kIndexAddOrderedWork
// index loads per cloud, float atomics into a zeroed destination beyond
template <bool CHANNEL_MAJOR>
static int launch_index_add(int b, int n, int c, long long ne, const int *idx, const float *src, float *dst, hipStream_t st)
{
    SN_REQUIRE(ne <= 0x7fffffffll, "m * nsample must fit in 31 bits");
    if ((long long)((n + 63) / 64) * ne <= kIndexAddOrderedWork) {
        hipLaunchKernelGGL(index_add_ordered_kernel<CHANNEL_MAJOR>, dim3((n + 255) / 256, b), dim3(256), 0, st, n, c, (int)ne, idx,
                           src, dst);
    } else {
        const hipError_t e = hipMemsetAsync(dst, 0, (size_t)b * n * c * sizeof(float), st);
        if (e != hipSuccess) return sn_set_error((int)e, "%s: %s", __func__, hipGetErrorString(e));
        hipLaunchKernelGGL(index_add_atomic_kernel<CHANNEL_MAJOR>, dim3(grid_for((size_t)ne * c), b), dim3(256), 0, st, n, c, ne,
                           idx, src, dst);
    }
    SN_LAUNCH_CHECK();
    return 0;
}

// The three backward entries above with the gradient towards the point cloud / the features summed in the ORDER of the
// reference's CPU loops (query ascending, neighbour ascending) instead of by unordered float atomics: the kernels store the
// per-(query, neighbour) contributions in `scratch` (b * m * k * 3 floats; b * c * m * k for the gather) and
// index_add_ordered_kernel adds, per destination row, the entries that name it in ascending entry order.  grad_P / grad_X are
// OVERWRITTEN (every row is written), bit-identical to oracle/samplenet_oracle.c: orc_softproj_backward, run to run.
extern "C" int sn_soft_project_backward_ordered(int b, int n, int m, int k, const float *P, int p_layout, const float *Q,
                                                int q_layout, const int *idx, const float *temperature, float min_sigma,
                                                const float *grad_proj, int gproj_layout, float *grad_Q, int gq_layout,
                                                float *grad_P, float *grad_sigma_partial, float *scratch, sn_stream_t stream)
{
    SN_REQUIRE(b >= 0 && n >= 1 && m >= 0 && k >= 1 && k <= 64, "bad size");
    if (b == 0 || m == 0) return 0;
    SN_REQUIRE(P && Q && idx && temperature && grad_proj && grad_P && scratch, "null pointer");
    SoftBwdArgs a{};
    a.P = P, a.Q = Q, a.idx = idx, a.temperature = temperature, a.min_sigma = min_sigma;
    a.p_layout = p_layout, a.q_layout = q_layout, a.n = n, a.m = m, a.k = k;
    a.grad_proj = grad_proj, a.gproj_layout = gproj_layout;
    a.grad_Q = grad_Q, a.gq_layout = gq_layout, a.grad_P = grad_P, a.gp_contrib = scratch, a.grad_sigma_partial = grad_sigma_partial;
    hipLaunchKernelGGL(soft_bwd_kernel<true>, dim3(b, sn_soft_bwd_splits(b, m)), dim3(256), 0, (hipStream_t)stream, a);
    return p_layout == SN_LAYOUT_BNC ? launch_index_add<false>(b, n, 3, (long long)m * k, idx, scratch, grad_P, (hipStream_t)stream)
                                     : launch_index_add<true>(b, n, 3, (long long)m * k, idx, scratch, grad_P, (hipStream_t)stream);
}

extern "C" int sn_soft_weights_backward_ordered(int b, int n, int m, int k, const float *P, const float *Q, const int *idx,
                                                const float *temperature, float min_sigma, const float *weights,
                                                const float *grad_weights, float *grad_Q, float *grad_P,
                                                float *grad_sigma_partial, float *scratch, sn_stream_t stream)
{
    SN_REQUIRE(b >= 0 && n >= 1 && m >= 0 && k >= 1 && k <= 64, "bad size");
    if (b == 0 || m == 0) return 0;
    SN_REQUIRE(P && Q && idx && temperature && grad_weights && grad_P && scratch, "null pointer");
    SoftBwdArgs a{};
    a.P = P, a.Q = Q, a.idx = idx, a.temperature = temperature, a.min_sigma = min_sigma;
    a.p_layout = SN_LAYOUT_BCN, a.q_layout = SN_LAYOUT_BCN, a.n = n, a.m = m, a.k = k;
    a.weights_in = weights, a.grad_weights = grad_weights;
    a.grad_Q = grad_Q, a.gq_layout = SN_LAYOUT_BCN, a.grad_P = grad_P, a.gp_contrib = scratch, a.grad_sigma_partial = grad_sigma_partial;
    hipLaunchKernelGGL(soft_bwd_kernel<false>, dim3(b, sn_soft_bwd_splits(b, m)), dim3(256), 0, (hipStream_t)stream, a);
    return launch_index_add<true>(b, n, 3, (long long)m * k, idx, scratch, grad_P, (hipStream_t)stream);
}

extern "C" int sn_weighted_gather_backward_ordered(int b, int c, int n, int m, int k, const float *X, const int *idx,
                                                   const float *weights, const float *grad_out, float *grad_weights,
                                                   float *grad_X, float *scratch, sn_stream_t stream)
{
    SN_REQUIRE(b >= 0 && c >= 0 && n >= 1 && m >= 0 && k >= 1, "bad size");
    if (b == 0 || m == 0) return 0;
    SN_REQUIRE(X && idx && weights && grad_out && grad_X && scratch, "null pointer");
    hipLaunchKernelGGL(weighted_gather_bwd_kernel, dim3(((size_t)m * k + 255) / 256, b), dim3(256), 0, (hipStream_t)stream, c, n, m,
                       k, X, idx, weights, grad_out, grad_weights, grad_X, scratch);
    if (c == 0) return 0;
    return launch_index_add<true>(b, n, c, (long long)m * k, idx, scratch, grad_X, (hipStream_t)stream);
}

extern "C" int sn_group_point(int b, int n, int c, int m, int nsample, const float *points, const int *idx,
                              float *out, sn_stream_t stream)
{
    SN_REQUIRE(b >= 0 && n >= 0 && c >= 0 && m >= 0 && nsample >= 0, "negative size");
    const size_t tot = (size_t)m * nsample * c;
    if (b == 0 || tot == 0) return 0;
    SN_REQUIRE(points && idx && out, "null pointer");
    hipLaunchKernelGGL(group_point_kernel, dim3(grid_for(tot), b), dim3(256), 0, (hipStream_t)stream, n, c, m,
                       nsample, points, idx, out);
    SN_LAUNCH_CHECK();
    return 0;
}

extern "C" int sn_group_point_grad(int b, int n, int c, int m, int nsample, const float *grad_out, const int *idx,
                                   float *grad_points, sn_stream_t stream)
{
    SN_REQUIRE(b >= 0 && n >= 0 && c >= 0 && m >= 0 && nsample >= 0, "negative size");
    if (b == 0 || (size_t)n * c == 0) return 0;
    SN_REQUIRE(grad_points, "null pointer");
    SN_REQUIRE((size_t)m * nsample == 0 || (grad_out && idx), "null pointer");
    return launch_index_add<false>(b, n, c, (long long)m * nsample, idx, grad_out, grad_points, (hipStream_t)stream);
}

extern "C" int sn_grouping_operation(int b, int c, int n, int m, int nsample, const float *features,
                                     const int *idx, float *out, sn_stream_t stream)
{
    SN_REQUIRE(b >= 0 && n >= 0 && c >= 0 && m >= 0 && nsample >= 0, "negative size");
    const size_t tot = (size_t)m * nsample * c;
    if (b == 0 || tot == 0) return 0;
    SN_REQUIRE(features && idx && out, "null pointer");
    hipLaunchKernelGGL(grouping_operation_kernel, dim3(grid_for(tot), b), dim3(256), 0, (hipStream_t)stream, c, n, m,
                       nsample, features, idx, out);
    SN_LAUNCH_CHECK();
    return 0;
}

extern "C" int sn_grouping_operation_grad(int b, int c, int n, int m, int nsample, const float *grad_out,
                                          const int *idx, float *grad_features, sn_stream_t stream)
{
    SN_REQUIRE(b >= 0 && n >= 0 && c >= 0 && m >= 0 && nsample >= 0, "negative size");
    if (b == 0 || (size_t)n * c == 0) return 0;
    SN_REQUIRE(grad_features, "null pointer");
    SN_REQUIRE((size_t)m * nsample == 0 || (grad_out && idx), "null pointer");
    return launch_index_add<true>(b, n, c, (long long)m * nsample, idx, grad_out, grad_features, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------------
// The sampler's training-step loss in as few launches as the data dependencies allow (engine fast path; the op-by-op
// route through sn_simplification_loss_* / sn_sampler_loss_* / sn_soft_project_backward computes the same numbers):
//   L = alpha * (mean dq + mean_b max_m dq + weight * mean dp) + lmbda * max(T^2, min_sigma) + mean(proj)
// forward : [per cloud: finish the per-point minima from the pair scan's G partial key sets -> dp / ip, and reduce
//            sum dq, max dq (+ first argmax), sum dp, sum proj]  ->  [combine the clouds in order -> L]
// backward: [Chamfer backward with implicit upstream gradients (scaled by alpha) -> grad_Q]
//           [soft-projection backward with the constant upstream gradient g / nproj, ADDED to grad_Q; sigma partials]
//           [sigma partials + the direct lmbda term -> grad_T]
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024) step_loss_partial_kernel(int M, int N, int G, int nproj, const float *__restrict__ dq,
                                                                 const sn_u64 *__restrict__ ws, const float *__restrict__ proj,
                                                                 float *__restrict__ dp, int *__restrict__ ip,
                                                                 float *__restrict__ part, int *__restrict__ argmax1)
{
    // one workgroup of 1024 threads per cloud: at N = 1024 every thread finishes one point (its G keys in flight together)
    constexpr int NT = 1024;
    __shared__ float r0[NT], r1[NT], r2[NT], r3[NT];
    __shared__ int ri[NT];
    const int b = blockIdx.x, t = threadIdx.x;
    float s1 = 0.f, mx = -INFINITY, s2 = 0.f, sp = 0.f;
    int am = 0;
    for (int j = t; j < M; j += NT) {
        const float v = dq[(size_t)b * M + j];
        s1 += v;
        if (v > mx) mx = v, am = j;
    }
    for (int n = t; n < N; n += NT) {
        sn_u64 k = kKeyInf;
        for (int g = 0; g < G; ++g) {  // minimum of (distance, query) keys = lowest query on ties
            const sn_u64 v = ws[((size_t)b * G + g) * N + n];
            k = v < k ? v : k;
        }
        const float d = key_dist(k);
        dp[(size_t)b * N + n] = d;
        ip[(size_t)b * N + n] = key_index(k);
        s2 += d;
    }
    for (int i = t; i < nproj; i += NT) sp += proj[(size_t)b * nproj + i];
    r0[t] = s1, r1[t] = mx, r2[t] = s2, r3[t] = sp, ri[t] = am;
    for (int s = NT / 2; s > 0; s >>= 1) {
        __syncthreads();
        if (t < s) {
            r0[t] += r0[t + s];
            r2[t] += r2[t + s];
            r3[t] += r3[t + s];
            if (r1[t + s] > r1[t] || (r1[t + s] == r1[t] && ri[t + s] < ri[t])) r1[t] = r1[t + s], ri[t] = ri[t + s];
        }
    }
    if (t == 0) {
        part[b * 4 + 0] = r0[0], part[b * 4 + 1] = r1[0], part[b * 4 + 2] = r2[0], part[b * 4 + 3] = r3[0];
        argmax1[b] = ri[0];
    }
}

__global__ void __launch_bounds__(64) step_loss_final_kernel(StepLossFinal f) { step_loss_final(f, threadIdx.x); }

extern "C" int sn_sampler_step_loss_forward(int B, int M, int N, int G, const float *dist_q, const void *colmin_ws,
                                            const float *proj, const float *temperature, float alpha, float lmbda,
                                            float weight, float min_sigma, float *dist_p, int *idx_p, int *argmax1,
                                            float *partial, float *loss, int defer_value, sn_stream_t stream)
{
    SN_REQUIRE(B >= 1 && M >= 1 && N >= 1 && G >= 1, "bad size");
    SN_REQUIRE(dist_q && colmin_ws && proj && temperature && dist_p && idx_p && argmax1 && partial && loss, "null pointer");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(step_loss_partial_kernel, dim3(B), dim3(1024), 0, st, M, N, G, 3 * M, dist_q, (const sn_u64 *)colmin_ws,
                       proj, dist_p, idx_p, partial, argmax1);
    // defer_value: the scalar is combined by an extra wave of sn_sampler_step_loss_backward's first launch (the gradients do
    // not need it; a launch of its own costs ~4.7 us inside the step's graph) -- only for callers that always run backward
    if (!defer_value) {
        const StepLossFinal f{B, M, N, 3 * M, weight, alpha, lmbda, min_sigma, partial, temperature, loss};
        hipLaunchKernelGGL(step_loss_final_kernel, dim3(1), dim3(64), 0, st, f);
    }
    SN_LAUNCH_CHECK();
    return 0;
}

// sn_sampler_step_loss_forward behind a scan that ran ONE workgroup per cloud (batches that fill the chip on their own:
// sn_pairscan_colmin_splits(B, N, M) <= 1 -- dist_p / idx_p are complete, there are no partial key sets): per-cloud sums over
// dist_q / dist_p / proj and the arg-max of dist_q, then the clouds in order.  Same partial layout, same final kernel.
__global__ void __launch_bounds__(256) step_loss_partial_direct_kernel(int M, int N, int nproj, const float *__restrict__ dq,
                                                                       const float *__restrict__ dp, const float *__restrict__ proj,
                                                                       float *__restrict__ part, int *__restrict__ argmax1)
{
    constexpr int NT = 256;
    __shared__ float r0[NT], r1[NT], r2[NT], r3[NT];
    __shared__ int ri[NT];
    const int b = blockIdx.x, t = threadIdx.x;
    float s1 = 0.f, mx = -INFINITY, s2 = 0.f, sp = 0.f;
    int am = 0;
    for (int j = t; j < M; j += NT) {
        const float v = dq[(size_t)b * M + j];
        s1 += v;
        if (v > mx) mx = v, am = j;
    }
    for (int n = t; n < N; n += NT) s2 += dp[(size_t)b * N + n];
    for (int i = t; i < nproj; i += NT) sp += proj[(size_t)b * nproj + i];
    r0[t] = s1, r1[t] = mx, r2[t] = s2, r3[t] = sp, ri[t] = am;
    for (int s = NT / 2; s > 0; s >>= 1) {
        __syncthreads();
        if (t < s) {
            r0[t] += r0[t + s];
            r2[t] += r2[t + s];
            r3[t] += r3[t + s];
            if (r1[t + s] > r1[t] || (r1[t + s] == r1[t] && ri[t + s] < ri[t])) r1[t] = r1[t + s], ri[t] = ri[t + s];
        }
    }
    if (t == 0) {
        part[b * 4 + 0] = r0[0], part[b * 4 + 1] = r1[0], part[b * 4 + 2] = r2[0], part[b * 4 + 3] = r3[0];
        argmax1[b] = ri[0];
    }
}

extern "C" int sn_sampler_step_loss_forward_direct(int B, int M, int N, const float *dist_q, const float *dist_p, const float *proj,
                                                   const float *temperature, float alpha, float lmbda, float weight,
                                                   float min_sigma, int *argmax1, float *partial, float *loss, int defer_value,
                                                   sn_stream_t stream)
{
    SN_REQUIRE(B >= 1 && M >= 1 && N >= 1, "bad size");
    SN_REQUIRE(dist_q && dist_p && proj && temperature && argmax1 && partial && loss, "null pointer");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(step_loss_partial_direct_kernel, dim3(B), dim3(256), 0, st, M, N, 3 * M, dist_q, dist_p, proj, partial, argmax1);
    if (!defer_value) {
        const StepLossFinal f{B, M, N, 3 * M, weight, alpha, lmbda, min_sigma, partial, temperature, loss};
        hipLaunchKernelGGL(step_loss_final_kernel, dim3(1), dim3(64), 0, st, f);
    }
    SN_LAUNCH_CHECK();
    return 0;
}

// grad_Q (B,3,M) channel-major (the FC head's layout), grad_T (1).  P: (B,N,3) or (B,3,N) by p_layout; Q: (B,3,M).
// gsig_scratch: B * sn_soft_bwd_splits(B, M) floats.
extern "C" int sn_sampler_step_loss_backward(int B, int N, int M, int K, const float *P, int p_layout, const float *Q,
                                             const int *knn_idx, const int *idx_q, const int *idx_p, const int *argmax1,
                                             const float *temperature, float min_sigma, float alpha, float lmbda,
                                             float weight, const float *grad_loss, float *grad_Q, float *gsig_scratch,
                                             float *grad_T, const float *deferred_partial, float *deferred_loss,
                                             sn_stream_t stream)
{
    SN_REQUIRE(B >= 1 && N >= 1 && M >= 1 && K >= 1 && K <= 64, "bad size");
    SN_REQUIRE(!deferred_loss || deferred_partial, "deferred loss value needs the forward's partials");
    SN_REQUIRE(p_layout == SN_LAYOUT_BNC, "the reference cloud must be (B,N,3) here");
    SN_REQUIRE(P && Q && knn_idx && idx_q && idx_p && argmax1 && temperature && grad_loss && grad_Q && gsig_scratch && grad_T,
               "null pointer");
    hipStream_t st = (hipStream_t)stream;
    // same roundings as the op-by-op route: (alpha * g) first, then the per-term coefficients
    const float c1 = 1.0f / ((float)B * (float)M), cm = 1.0f / (float)B, c2 = weight / ((float)B * (float)N);
    ImplicitGrad ig{grad_loss, argmax1, nullptr, c1, cm, c2, 0.f, alpha};
    SoftBwdArgs a{};
    a.P = P, a.Q = Q, a.idx = knn_idx, a.temperature = temperature, a.min_sigma = min_sigma;
    a.p_layout = p_layout, a.q_layout = SN_LAYOUT_BCN, a.n = N, a.m = M, a.k = K;
    a.grad_proj = nullptr, a.gconst = grad_loss, a.gconst_div = (float)(B * 3 * M);
    a.grad_Q = grad_Q, a.gq_layout = SN_LAYOUT_BCN, a.accumulate_q = 1;
    a.grad_P = nullptr, a.grad_sigma_partial = gsig_scratch;
    int splits = sn_soft_bwd_splits(B, M);
    if (N <= 2048) {
        // 1+2. alpha * d L_simp / d Q  +  d mean(proj) / d Q  (and the sigma partials) in one launch
        splits = std::max(1, std::min((M + 3) / 4, (kChamferBwdGroups + B - 1) / B));
        splits = std::min(splits, sn_soft_bwd_splits(B, M));  // gsig_scratch is sized by the caller for that many
        const StepLossFinal fin{B, M, N, 3 * M, weight, alpha, lmbda, min_sigma, deferred_partial, temperature, deferred_loss};
        const dim3 grid(B, splits + (deferred_loss ? 1 : 0)), block(256);
#define SN_CS(PPL_) \
    hipLaunchKernelGGL(chamfer_soft_bwd_kernel<PPL_>, grid, block, 0, st, M, N, Q, P, idx_q, idx_p, grad_Q, ig, a, fin, StepLossFold{})
        if (N <= 64) SN_CS(1);
        else if (N <= 256) SN_CS(4);
        else if (N <= 1024) SN_CS(16);
        else SN_CS(32);
#undef SN_CS
    } else {
        // 1. alpha * d L_simp / d Q   (targets = Q, channel-major; sources = P)
        auto ysplit = [&](int nt) { return std::max(1, std::min((nt + 3) / 4, (kChamferBwdGroups + B - 1) / B)); };
        launch_chamfer_bwd(B, ysplit(M), M, N, Q, P, nullptr, idx_q, nullptr, idx_p, grad_Q, 1, ig, st, 1);
        // 2. + d mean(proj) / d Q, and the sigma partials
        hipLaunchKernelGGL(soft_bwd_kernel<true>, dim3(B, splits), dim3(256), 0, st, a);
        if (deferred_loss) {
            const StepLossFinal fin{B, M, N, 3 * M, weight, alpha, lmbda, min_sigma, deferred_partial, temperature, deferred_loss};
            hipLaunchKernelGGL(step_loss_final_kernel, dim3(1), dim3(64), 0, st, fin);
        }
    }
    // 3. grad_T from the sigma partials and the direct lmbda * sigma term
    hipLaunchKernelGGL(sigma_grad_kernel, dim3(1), dim3(256), 0, st, B * splits, gsig_scratch, temperature, min_sigma, grad_T,
                       grad_loss, lmbda);
    SN_LAUNCH_CHECK();
    return 0;
}

// Loss side of the sampler step behind sn_pairscan_forward_keys (engine): backward + loss value in 2 launches, nothing
// between the scan and the backward.  colmin_keys / qpart / qmax as the scan left them; colmin_keys is zero again afterwards.
// dpsum: B floats of scratch; loss[0] = L, loss[1] = L_simp.  N <= 2048.
extern "C" int sn_sampler_step_loss_keys(int B, int N, int M, int K, const float *P, int p_layout, const float *Q,
                                         const int *knn_idx, const int *idx_q, void *colmin_keys, const float *qpart,
                                         const void *qmax, int G, const float *temperature, float min_sigma, float alpha,
                                         float lmbda, float weight, const float *grad_loss, float *grad_Q,
                                         float *gsig_scratch, float *grad_T, float *dpsum, float *loss, sn_stream_t stream,
                                         void *deferred_tail, const float *grad_proj, const float *grad_sigma)
{
    SN_REQUIRE(B >= 1 && N >= 1 && M >= 1 && K >= 1 && K <= 64 && G >= 1, "bad size");
    SN_REQUIRE(p_layout == SN_LAYOUT_BNC, "the reference cloud must be (B,N,3) here");
    SN_REQUIRE(P && Q && knn_idx && idx_q && colmin_keys && qpart && qmax && temperature && grad_loss && grad_Q && gsig_scratch &&
                   grad_T && dpsum && loss,
               "null pointer");
    if (N > 2048) return sn_set_error(SN_ERR_UNSUPPORTED, "sn_sampler_step_loss_keys: N <= 2048");
    hipStream_t st = (hipStream_t)stream;
    const float c1 = 1.0f / ((float)B * (float)M), cm = 1.0f / (float)B, c2 = weight / ((float)B * (float)N);
    ImplicitGrad ig{grad_loss, nullptr, nullptr, c1, cm, c2, 0.f, alpha};
    SoftBwdArgs a{};
    a.P = P, a.Q = Q, a.idx = knn_idx, a.temperature = temperature, a.min_sigma = min_sigma;
    a.p_layout = p_layout, a.q_layout = SN_LAYOUT_BCN, a.n = N, a.m = M, a.k = K;
    // grad_proj (B,M,3), optional: the task loss's gradient w.r.t. the projected points (main.py:507-531: the task network
    // sits on proj); NULL = the benchmark's stand-in term mean(proj) inside this loss, upstream gradient grad_loss / (3BM)
    a.grad_proj = grad_proj, a.gproj_layout = SN_LAYOUT_BNC, a.gconst = grad_loss, a.gconst_div = (float)(B * 3 * M);
    a.grad_Q = grad_Q, a.gq_layout = SN_LAYOUT_BCN, a.accumulate_q = 1;
    a.grad_P = nullptr, a.grad_sigma_partial = gsig_scratch;
    int splits = std::max(1, std::min((M + 3) / 4, (kChamferBwdGroups + B - 1) / B));
    splits = std::min(splits, sn_soft_bwd_splits(B, M));
    StepLossFold fold{};
    fold.G = G, fold.keys = (const sn_u64 *)colmin_keys, fold.qmax = (const sn_u64 *)qmax, fold.dpsum = dpsum;
    const dim3 grid(B, splits), block(256);
#define SN_CS(PPL_)                                                                                                        \
    hipLaunchKernelGGL(chamfer_soft_bwd_kernel<PPL_>, grid, block, 0, st, M, N, Q, P, idx_q, (const int *)nullptr, grad_Q, ig, a, \
                       StepLossFinal{}, fold)
    if (N <= 64) SN_CS(1);
    else if (N <= 256) SN_CS(4);
    else if (N <= 1024) SN_CS(16);
    else SN_CS(32);
#undef SN_CS
    const StepLossKeysFinal kf{B, G, M, N, 3 * M, weight, alpha, lmbda, min_sigma, qpart, (const sn_u64 *)qmax, dpsum, temperature,
                               loss, (sn_u64 *)colmin_keys, (long long)B * N, grad_proj ? 0 : 1, {nullptr, nullptr}};
    if (deferred_tail) {
        // the caller hands this blob to a later launch of the same step (sn_conv_stack_backward runs it in two extra
        // workgroups of its closing kernel): no launch of its own for the sigma gradient / loss value / key reset
        StepTail t{};
        t.nparts = B * splits, t.gsig = gsig_scratch, t.temperature = temperature, t.min_sigma = min_sigma, t.grad_T = grad_T;
        // (grad_sigma: sigma is an output of the caller's node with an upstream gradient of its own -- the drop-in surface,
        //  where the script forms lmbda * get_projection_loss() itself; else the direct term is lmbda * grad_loss)
        t.grad_loss = grad_sigma ? grad_sigma : grad_loss, t.lmbda = grad_sigma ? 1.f : lmbda, t.kf = kf;
        memcpy(deferred_tail, &t, sizeof(t));
        SN_LAUNCH_CHECK();
        return 0;
    }
    hipLaunchKernelGGL(sigma_grad_kernel, dim3(2 + 64), dim3(256), 0, st, B * splits, gsig_scratch, temperature, min_sigma, grad_T,
                       grad_sigma ? grad_sigma : grad_loss, grad_sigma ? 1.f : lmbda, StepLossFinal{}, kf);
    SN_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// The drop-in module surface on captured work (samplenet_amd/surface.py): registration/main.py:507-531 calls net(x), then
// get_simplification_loss() / get_projection_loss(), weights them itself and calls backward().  The forward graph therefore
// needs the VALUES of L_simp and sigma right behind the keys-mode scan (the engine's fused step gets them from the backward's
// tail), and the backward graph takes three upstream gradients (d / d L_simp, d / d sigma, d / d proj) from device memory.
//   sn_surface_values_keys:  [per cloud: sum of the per-point minima out of the key table (same order as the loss backward's
//                            own sum: strided per-thread sums, xor tree, waves in order); simplified cloud (B,3,M) -> (B,M,3)]
//                            -> [one wave: the clouds in the order of step_loss_keys_final]   values[0] = L_simp,
//                            values[1] = sigma, values[2..4] = mean dist_q, mean_b max dist_q, mean dist_p
//   sn_surface_gather_upstream: the three upstream gradients into the static operands of the captured backward (an absent one
//                            becomes zero) -- one launch instead of three copies
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) surface_cloud_values_kernel(int N, int M, const sn_u64 *__restrict__ keys,
                                                                   const float *__restrict__ y_bcn, float *__restrict__ simp_bnc,
                                                                   float *__restrict__ dpsum)
{
    __shared__ float red[4];
    const int b = blockIdx.x, t = threadIdx.x;
    float sdp = 0.f;
    for (int n = t; n < N; n += 256) sdp += key_dist(~keys[(size_t)b * N + n]);
    if (simp_bnc)
        for (int i = t; i < 3 * M; i += 256) simp_bnc[(size_t)b * 3 * M + i] = y_bcn[(size_t)b * 3 * M + (i % 3) * M + i / 3];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sdp += __shfl_xor(sdp, o);
    if ((t & 63) == 0) red[t >> 6] = sdp;
    __syncthreads();
    if (t == 0) dpsum[b] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ void __launch_bounds__(64) surface_values_final_kernel(int B, int G, int M, int N, float w, float min_sigma,
                                                                  const float *__restrict__ qpart, const sn_u64 *__restrict__ qmax,
                                                                  const float *__restrict__ dpsum,
                                                                  const float *__restrict__ temperature, float *__restrict__ values)
{
    const int t = threadIdx.x;
    float s1 = 0.f, mx = 0.f, s2 = 0.f;
    for (int b = t; b < B; b += 64) {
        float a1 = 0.f;
        sn_u64 mk = 0;
        for (int g = 0; g < G; ++g) {
            const size_t o = (size_t)b * G + g;
            a1 += qpart[o * 2];
            mk = qmax[o] > mk ? qmax[o] : mk;
        }
        s1 += a1, mx += key_dist(mk), s2 += dpsum[b];
    }
    const float T = *temperature;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        s1 += __shfl_xor(s1, o);
        mx += __shfl_xor(mx, o);
        s2 += __shfl_xor(s2, o);
    }
    if (t != 0) return;
    const float c12 = s1 / ((float)B * (float)M), cmax = mx / (float)B, c21 = s2 / ((float)B * (float)N);
    values[0] = c12 + cmax + w * c21;
    values[1] = sn_sigma(T, min_sigma);
    values[2] = c12, values[3] = cmax, values[4] = c21;
}

extern "C" int sn_surface_values_keys(int B, int N, int M, int G, const void *colmin_keys, const float *qpart, const void *qmax,
                                      const float *temperature, float min_sigma, float weight, const float *y_bcn,
                                      float *simp_bnc, float *dpsum, float *values, sn_stream_t stream)
{
    SN_REQUIRE(B >= 1 && N >= 1 && M >= 1 && G >= 1, "bad size");
    SN_REQUIRE(colmin_keys && qpart && qmax && temperature && dpsum && values && (y_bcn || !simp_bnc), "null pointer");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(surface_cloud_values_kernel, dim3(B), dim3(256), 0, st, N, M, (const sn_u64 *)colmin_keys, y_bcn, simp_bnc,
                       dpsum);
    hipLaunchKernelGGL(surface_values_final_kernel, dim3(1), dim3(64), 0, st, B, G, M, N, weight, min_sigma, qpart,
                       (const sn_u64 *)qmax, dpsum, temperature, values);
    SN_LAUNCH_CHECK();
    return 0;
}

__global__ void __launch_bounds__(256) surface_gather_upstream_kernel(int nproj, const float *__restrict__ g_lsimp,
                                                                      const float *__restrict__ g_sigma,
                                                                      const float *__restrict__ g_proj, float *__restrict__ scalars,
                                                                      float *__restrict__ proj_out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < nproj) proj_out[i] = g_proj ? g_proj[i] : 0.f;
    if (i == 0) scalars[0] = g_lsimp ? *g_lsimp : 0.f, scalars[1] = g_sigma ? *g_sigma : 0.f;
}

extern "C" int sn_surface_gather_upstream(int nproj, const float *g_lsimp, const float *g_sigma, const float *g_proj,
                                          float *scalars, float *proj_out, sn_stream_t stream)
{
    SN_REQUIRE(nproj >= 1 && scalars && proj_out, "bad argument");
    hipLaunchKernelGGL(surface_gather_upstream_kernel, dim3((nproj + 255) / 256), dim3(256), 0, (hipStream_t)stream, nproj,
                       g_lsimp, g_sigma, g_proj, scalars, proj_out);
    SN_LAUNCH_CHECK();
    return 0;
}

extern "C" int sn_step_tail_bytes(void) { return (int)sizeof(sn::StepTail); }

// Names the error words of the step's FC chain launches (sn_fc_chain_forward* / sn_fc_chain_backward: word 15 of their sync
// buffers; either may be NULL) in a tail blob: the deferred loss value becomes NaN when one of them is set.
extern "C" int sn_step_tail_set_error_words(void *tail_blob, const void *fwd_sync, const void *bwd_sync)
{
    SN_REQUIRE(tail_blob, "null blob");
    StepTail t;
    memcpy(&t, tail_blob, sizeof(t));
    t.kf.chain_err[0] = fwd_sync ? (const unsigned *)fwd_sync + 15 * SN_FC_SYNC_STRIDE : nullptr;
    t.kf.chain_err[1] = bwd_sync ? (const unsigned *)bwd_sync + 15 * SN_FC_SYNC_STRIDE : nullptr;
    memcpy(tail_blob, &t, sizeof(t));
    return 0;
}

// ------------------------------------------------------------------------------------------------
// Rotation of a cloud by one quaternion per cloud (the registration task loss rotates the template by the
// estimated pose -- registration/main.py:569-571 est_transform.rotate(p0) = src/quaternion.py:35-53 qrot with the quaternion
// expanded over the points).  The reference composes it from two cross products, a scale and two adds per point on
// materialised (B,N,4) / (B,N,3) tensors: ~8 launches forward, ~16 backward.  Here: one launch each way.
//   out = v + 2 (w (u x v) + u x (u x v)),   q = (w, u)
//   backward:  dv = g - 2 w (u x g) + 2 u x (u x g)                                   (rotation by the conjugate)
//              dw = sum_n 2 (u x v_n) . g_n
//              du = sum_n 2 w (v_n x g_n) + 2 (u . g_n) v_n + 2 (u . v_n) g_n - 4 (v_n . g_n) u
// one workgroup per cloud; the per-cloud sums in a fixed order (strided per-thread partials, xor tree, waves in order).
// ------------------------------------------------------------------------------------------------
struct sn_v3 {
    float x, y, z;
};
__device__ __forceinline__ sn_v3 cross3(sn_v3 a, sn_v3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ __forceinline__ float dot3(sn_v3 a, sn_v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }

__global__ void __launch_bounds__(256) qrot_fwd_kernel(int N, const float *__restrict__ q, const float *__restrict__ v,
                                                       float *__restrict__ out)
{
    const int b = blockIdx.y;
    const float w = q[b * 4];
    const sn_v3 u{q[b * 4 + 1], q[b * 4 + 2], q[b * 4 + 3]};
    for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < N; n += gridDim.x * blockDim.x) {
        const sn_xyz3 p = *reinterpret_cast<const sn_xyz3 *>(v + ((size_t)b * N + n) * 3);
        const sn_v3 x{p.x, p.y, p.z};
        const sn_v3 uv = cross3(u, x), uuv = cross3(u, uv);
        sn_xyz3 o;
        o.x = x.x + 2.0f * (w * uv.x + uuv.x), o.y = x.y + 2.0f * (w * uv.y + uuv.y), o.z = x.z + 2.0f * (w * uv.z + uuv.z);
        *reinterpret_cast<sn_xyz3 *>(out + ((size_t)b * N + n) * 3) = o;
    }
}

__global__ void __launch_bounds__(256) qrot_bwd_kernel(int N, const float *__restrict__ q, const float *__restrict__ v,
                                                       const float *__restrict__ g, float *__restrict__ gq,
                                                       float *__restrict__ gv)
{
    __shared__ float red[4][4];
    const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float w = q[b * 4];
    const sn_v3 u{q[b * 4 + 1], q[b * 4 + 2], q[b * 4 + 3]};
    float aw = 0.f, ax = 0.f, ay = 0.f, az = 0.f;
    for (int n = threadIdx.x; n < N; n += 256) {
        const sn_xyz3 gp = *reinterpret_cast<const sn_xyz3 *>(g + ((size_t)b * N + n) * 3);
        const sn_v3 gg{gp.x, gp.y, gp.z};
        if (gv) {
            const sn_v3 ug = cross3(u, gg), uug = cross3(u, ug);
            sn_xyz3 o;
            o.x = gg.x + 2.0f * (uug.x - w * ug.x), o.y = gg.y + 2.0f * (uug.y - w * ug.y), o.z = gg.z + 2.0f * (uug.z - w * ug.z);
            *reinterpret_cast<sn_xyz3 *>(gv + ((size_t)b * N + n) * 3) = o;
        }
        if (gq) {
            const sn_xyz3 p = *reinterpret_cast<const sn_xyz3 *>(v + ((size_t)b * N + n) * 3);
            const sn_v3 x{p.x, p.y, p.z};
            const sn_v3 uv = cross3(u, x), vg = cross3(x, gg);
            const float ug = dot3(u, gg), ux = dot3(u, x), xg = dot3(x, gg);
            aw += 2.0f * dot3(uv, gg);
            ax += 2.0f * (w * vg.x + ug * x.x + ux * gg.x) - 4.0f * xg * u.x;
            ay += 2.0f * (w * vg.y + ug * x.y + ux * gg.y) - 4.0f * xg * u.y;
            az += 2.0f * (w * vg.z + ug * x.z + ux * gg.z) - 4.0f * xg * u.z;
        }
    }
    if (!gq) return;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        aw += __shfl_xor(aw, o), ax += __shfl_xor(ax, o), ay += __shfl_xor(ay, o), az += __shfl_xor(az, o);
    }
    if (lane == 0) red[wave][0] = aw, red[wave][1] = ax, red[wave][2] = ay, red[wave][3] = az;
    __syncthreads();
    if (threadIdx.x < 4) gq[b * 4 + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// PCRNet's output head AND the rotation of the template by the estimated quaternion as ONE launch each way (the registration
// task term, registration/main.py:557-577: twist = model(p0, p1); p1_est = rotate(p0, twist[:, 0:4])): the arithmetic of
// pcrnet_head_fwd_kernel + qrot_fwd_kernel / qrot_bwd_kernel + pcrnet_head_bwd_kernel, expression for expression (results are
// bit-identical to the two launches), one dependent launch less in either direction of a latency-bound step.
//   forward : grid (x, B); every workgroup normalises its cloud's quaternion from y[b] itself and rotates its share of the points;
//             workgroup (0, b) stores twist[b] / quat[b]; workgroup (0, 0) also the regulariser (fixed-order sum over the batch).
//   backward: one workgroup per cloud: gv (optional), the quaternion's gradient by the fixed-order reduction of qrot_bwd_kernel,
//             then the head's backward of row b with that gradient in the place of g_quat.
// group > 0 (several evaluations against ONE template, the progressive sampler's prefixes): rows [e group, (e + 1) group) are evaluation e,
// v holds `group` clouds (row b rotates v[b % group]) and qnorm / g_qnorm hold one value per evaluation; 0: one evaluation of B rows.
__global__ void __launch_bounds__(256) pcrnet_head_rot_fwd_kernel(int B, int N, const float *__restrict__ y, const float *__restrict__ v,
                                                                  float *__restrict__ twist, float *__restrict__ quat,
                                                                  float *__restrict__ qnorm, float *__restrict__ out, int group)
{
    __shared__ float red[256];
    const int b = blockIdx.y;
    const int vb = group ? b % group : b;
    const float *r = y + (size_t)b * 7;
    const float n2 = r[0] * r[0] + r[1] * r[1] + r[2] * r[2] + r[3] * r[3];
    const float inv = 1.0f / fmaxf(sqrtf(n2), 1e-12f);
    const float q0 = r[0] * inv, q1 = r[1] * inv, q2 = r[2] * inv, q3 = r[3] * inv;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        float *o = twist + (size_t)b * 7;
        o[0] = q0, o[1] = q1, o[2] = q2, o[3] = q3, o[4] = r[4], o[5] = r[5], o[6] = r[6];
        quat[b * 4 + 0] = q0, quat[b * 4 + 1] = q1, quat[b * 4 + 2] = q2, quat[b * 4 + 3] = q3;
    }
    const float w = q0;
    const sn_v3 u{q1, q2, q3};
    for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < N; n += gridDim.x * blockDim.x) {
        const sn_xyz3 p = *reinterpret_cast<const sn_xyz3 *>(v + ((size_t)vb * N + n) * 3);
        const sn_v3 x{p.x, p.y, p.z};
        const sn_v3 uv = cross3(u, x), uuv = cross3(u, uv);
        sn_xyz3 o;
        o.x = x.x + 2.0f * (w * uv.x + uuv.x), o.y = x.y + 2.0f * (w * uv.y + uuv.y), o.z = x.z + 2.0f * (w * uv.z + uuv.z);
        *reinterpret_cast<sn_xyz3 *>(out + ((size_t)b * N + n) * 3) = o;
    }
    if (blockIdx.x != 0 || vb != 0 || !qnorm) return;  // (the first row of an evaluation: its regulariser)
    const int base = b;
    if (group) B = group, qnorm += b / group;
    float acc = 0.f;  // (pcrnet_head_fwd_kernel's sum: strided per-thread partials, halving tree)
    for (int bb = threadIdx.x; bb < B; bb += 256) {
        const float *rr = y + (size_t)(base + bb) * 7;
        const float m2 = rr[0] * rr[0] + rr[1] * rr[1] + rr[2] * rr[2] + rr[3] * rr[3];
        acc += (m2 - 1.0f) * (m2 - 1.0f);
    }
    red[threadIdx.x] = acc;
    for (int s = 128; s > 0; s >>= 1) {
        __syncthreads();
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    }
    if (threadIdx.x == 0) qnorm[0] = red[0] / (float)B;
}

__global__ void __launch_bounds__(256) pcrnet_head_rot_bwd_kernel(int B, int N, const float *__restrict__ y, const float *__restrict__ q,
                                                                  const float *__restrict__ v, const float *__restrict__ g,
                                                                  const float *__restrict__ g_twist, const float *__restrict__ g_quat,
                                                                  const float *__restrict__ g_qnorm, float *__restrict__ gv,
                                                                  float *__restrict__ g_y, int group)
{
    __shared__ float red[4][4];
    const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    v += (long long)((group ? b % group : b) - b) * N * 3;  // (v[b] below = the template of row b)
    if (group) B = group;
    if (group && g_qnorm) g_qnorm += b / group;
    const float w = q[b * 4];
    const sn_v3 u{q[b * 4 + 1], q[b * 4 + 2], q[b * 4 + 3]};
    float aw = 0.f, ax = 0.f, ay = 0.f, az = 0.f;
    if (g)
        for (int n = threadIdx.x; n < N; n += 256) {
            const sn_xyz3 gp = *reinterpret_cast<const sn_xyz3 *>(g + ((size_t)b * N + n) * 3);
            const sn_v3 gg{gp.x, gp.y, gp.z};
            if (gv) {
                const sn_v3 ug = cross3(u, gg), uug = cross3(u, ug);
                sn_xyz3 o;
                o.x = gg.x + 2.0f * (uug.x - w * ug.x), o.y = gg.y + 2.0f * (uug.y - w * ug.y), o.z = gg.z + 2.0f * (uug.z - w * ug.z);
                *reinterpret_cast<sn_xyz3 *>(gv + ((size_t)b * N + n) * 3) = o;
            }
            const sn_xyz3 p = *reinterpret_cast<const sn_xyz3 *>(v + ((size_t)b * N + n) * 3);
            const sn_v3 x{p.x, p.y, p.z};
            const sn_v3 uv = cross3(u, x), vg = cross3(x, gg);
            const float ug = dot3(u, gg), ux = dot3(u, x), xg = dot3(x, gg);
            aw += 2.0f * dot3(uv, gg);
            ax += 2.0f * (w * vg.x + ug * x.x + ux * gg.x) - 4.0f * xg * u.x;
            ay += 2.0f * (w * vg.y + ug * x.y + ux * gg.y) - 4.0f * xg * u.y;
            az += 2.0f * (w * vg.z + ug * x.z + ux * gg.z) - 4.0f * xg * u.z;
        }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        aw += __shfl_xor(aw, o), ax += __shfl_xor(ax, o), ay += __shfl_xor(ay, o), az += __shfl_xor(az, o);
    }
    if (lane == 0) red[wave][0] = aw, red[wave][1] = ax, red[wave][2] = ay, red[wave][3] = az;
    __syncthreads();
    if (threadIdx.x != 0) return;
    float gq4[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) gq4[i] = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
    // ---- the head's backward of row b (pcrnet_head_bwd_kernel) with gq4 (+ an explicit g_quat) as the normalised quaternion's gradient
    const float gqn = g_qnorm ? g_qnorm[0] : 0.f;
    const float *r = y + (size_t)b * 7;
    const float n2 = r[0] * r[0] + r[1] * r[1] + r[2] * r[2] + r[3] * r[3];
    const float nrm = sqrtf(n2);
    float *o = g_y + (size_t)b * 7;
    float gt[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) gt[i] = (g_twist ? g_twist[(size_t)b * 7 + i] : 0.f) + ((g ? gq4[i] : 0.f) + (g_quat ? g_quat[b * 4 + i] : 0.f));
    float gp[4];
    if (nrm > 1e-12f) {
        const float inv = 1.0f / nrm;
        const float qq[4] = {r[0] * inv, r[1] * inv, r[2] * inv, r[3] * inv};
        const float dot = qq[0] * gt[0] + qq[1] * gt[1] + qq[2] * gt[2] + qq[3] * gt[3];
#pragma unroll
        for (int i = 0; i < 4; ++i) gp[i] = (gt[i] - qq[i] * dot) * inv;
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) gp[i] = gt[i] * 1e12f;
    }
#pragma unroll
    for (int i = 4; i < 7; ++i) o[i] = g_twist ? g_twist[(size_t)b * 7 + i] : 0.f;
    const float cq = gqn * 4.0f * (n2 - 1.0f) / (float)B;
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = gp[i] + cq * r[i];
}

// y (B,7), v (B,N,3) -> twist (B,7), quat (B,4), qnorm (scalar, may be NULL), out (B,N,3) = v rotated by quat.
extern "C" int sn_pcrnet_head_rot_forward(int B, int N, const float *y, const float *v, float *twist, float *quat, float *qnorm,
                                          float *out, sn_stream_t stream)
{
    SN_REQUIRE(B >= 1 && N >= 1 && y && v && twist && quat && out, "bad argument");
    hipLaunchKernelGGL(pcrnet_head_rot_fwd_kernel, dim3(std::min((N + 255) / 256, 64), B), dim3(256), 0, (hipStream_t)stream, B, N, y, v,
                       twist, quat, qnorm, out, 0);
    SN_LAUNCH_CHECK();
    return 0;
}
// R = E * group rows: E evaluations against the SAME `group` template clouds v (group, N, 3); qnorm: E values (may be NULL)
extern "C" int sn_pcrnet_head_rot_forward_grouped(int R, int N, int group, const float *y, const float *v, float *twist, float *quat,
                                                  float *qnorm, float *out, sn_stream_t stream)
{
    SN_REQUIRE(R >= 1 && N >= 1 && group >= 1 && R % group == 0 && y && v && twist && quat && out, "bad argument");
    hipLaunchKernelGGL(pcrnet_head_rot_fwd_kernel, dim3(std::min((N + 255) / 256, 64), R), dim3(256), 0, (hipStream_t)stream, R, N, y, v,
                       twist, quat, qnorm, out, group);
    SN_LAUNCH_CHECK();
    return 0;
}

// Backward of sn_pcrnet_head_rot_forward: grad_out (B,N,3) (NULL: no gradient came through the rotated cloud), grad_twist (B,7),
// grad_quat (B,4), grad_qnorm (scalar) -- each may be NULL -- -> grad_y (B,7) and, when grad_v != NULL, grad_v (B,N,3) (needs grad_out).
extern "C" int sn_pcrnet_head_rot_backward(int B, int N, const float *y, const float *quat, const float *v, const float *grad_out,
                                           const float *grad_twist, const float *grad_quat, const float *grad_qnorm, float *grad_v,
                                           float *grad_y, sn_stream_t stream)
{
    SN_REQUIRE(B >= 1 && N >= 1 && y && quat && v && grad_y, "bad argument");
    SN_REQUIRE(!grad_v || grad_out, "grad_v needs grad_out");
    hipLaunchKernelGGL(pcrnet_head_rot_bwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, B, N, y, quat, v, grad_out, grad_twist,
                       grad_quat, grad_qnorm, grad_v, grad_y, 0);
    SN_LAUNCH_CHECK();
    return 0;
}
// backward of the grouped form: grad_qnorm E values (may be NULL); no gradient to the shared template clouds
extern "C" int sn_pcrnet_head_rot_backward_grouped(int R, int N, int group, const float *y, const float *quat, const float *v,
                                                   const float *grad_out, const float *grad_twist, const float *grad_quat,
                                                   const float *grad_qnorm, float *grad_y, sn_stream_t stream)
{
    SN_REQUIRE(R >= 1 && N >= 1 && group >= 1 && R % group == 0 && y && quat && v && grad_y, "bad argument");
    hipLaunchKernelGGL(pcrnet_head_rot_bwd_kernel, dim3(R), dim3(256), 0, (hipStream_t)stream, R, N, y, quat, v, grad_out, grad_twist,
                       grad_quat, grad_qnorm, (float *)nullptr, grad_y, group);
    SN_LAUNCH_CHECK();
    return 0;
}

extern "C" int sn_qrot_forward(int B, int N, const float *quat, const float *v, float *out, sn_stream_t stream)
{
    SN_REQUIRE(B >= 0 && N >= 0, "negative size");
    if (B == 0 || N == 0) return 0;
    SN_REQUIRE(quat && v && out, "null pointer");
    hipLaunchKernelGGL(qrot_fwd_kernel, dim3(std::min((N + 255) / 256, 64), B), dim3(256), 0, (hipStream_t)stream, N, quat, v, out);
    SN_LAUNCH_CHECK();
    return 0;
}

extern "C" int sn_qrot_backward(int B, int N, const float *quat, const float *v, const float *grad_out, float *grad_quat,
                                float *grad_v, sn_stream_t stream)
{
    SN_REQUIRE(B >= 0 && N >= 0, "negative size");
    if (B == 0) return 0;
    SN_REQUIRE(quat && v && grad_out && (grad_quat || grad_v), "null pointer");
    hipLaunchKernelGGL(qrot_bwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, N, quat, v, grad_out, grad_quat, grad_v);
    SN_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// Inference matching on the device: sputils.nn_matching (registration/src/sputils.py:7-41).
//   idx (B,k): nearest input point of every generated point.  complete_fps: keep the first occurrences in order
//   (np.unique(return_index) + sort), then farthest-point-complete to k points of the SAME cloud -- numpy computes the
//   distances in float64 ((p0 - points)**2).sum(1), argmax takes the first maximum; reproduced exactly: fp64
//   ((dx*dx + dy*dy) + dz*dz) without contraction, ties -> lowest point index.
// One workgroup of 1024 threads per cloud, PPT points per thread with their running minimum distance in registers; a
// farthest-point step = wave max (DPP/shuffle) -> 16 wave results through LDS -> every thread reads the winner.
// ------------------------------------------------------------------------------------------------
template <int PPT>
__global__ void __launch_bounds__(1024) nn_matching_kernel(int N, int K, int layout, int complete_fps,
                                                           const float *__restrict__ xyz, const int *__restrict__ idx,
                                                           float *__restrict__ out)
{
    constexpr int NT = 1024;
    __shared__ int s_sel[1024];      // selected point indices, first-occurrence order then FPS picks (K <= 1024)
    __shared__ int s_first[1024];    // 1 if idx[j] is a first occurrence
    __shared__ double s_wmax[16];
    __shared__ int s_warg[16];
    __shared__ int s_t;
    const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const float *P = xyz + (size_t)b * 3 * N;
    const int *id = idx + (size_t)b * K;
    float *o = out + (size_t)b * K * 3;

    if (!complete_fps) {
        for (int j = t; j < K; j += NT) {
            const int p = id[j];
#pragma unroll
            for (int c = 0; c < 3; ++c) o[j * 3 + c] = P[pt_off(layout, N, p, c)];
        }
        return;
    }
    // ---- unique, first-occurrence order
    for (int j = t; j < K; j += NT) {
        const int v = id[j];
        int first = 1;
        for (int q = 0; q < j; ++q) first &= (id[q] != v);
        s_first[j] = first;
    }
    __syncthreads();
    if (t == 0) {  // K is small (64): a sequential compaction keeps the order trivially right
        int n = 0;
        for (int j = 0; j < K; ++j)
            if (s_first[j]) s_sel[n++] = id[j];
        s_t = n;
    }
    __syncthreads();
    const int nseed = s_t;
    // ---- this thread's points and their distance to the seed set
    double px[PPT], py[PPT], pz[PPT], dist[PPT];
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
        const int p = min(t + i * NT, N - 1);
        px[i] = (double)P[pt_off(layout, N, p, 0)];
        py[i] = (double)P[pt_off(layout, N, p, 1)];
        pz[i] = (double)P[pt_off(layout, N, p, 2)];
        dist[i] = INFINITY;
    }
    for (int q = 0; q < nseed; ++q) {
        const int sp = s_sel[q];
        const double sx = (double)P[pt_off(layout, N, sp, 0)], sy = (double)P[pt_off(layout, N, sp, 1)],
                     sz = (double)P[pt_off(layout, N, sp, 2)];
#pragma unroll
        for (int i = 0; i < PPT; ++i) {
            const double dx = sx - px[i], dy = sy - py[i], dz = sz - pz[i];
            const double d = (dx * dx + dy * dy) + dz * dz;
            dist[i] = d < dist[i] ? d : dist[i];
        }
    }
    // ---- farthest-point completion
    for (int j = nseed; j < K; ++j) {
        double best = -1.0;
        int arg = 0x7fffffff;
#pragma unroll
        for (int i = 0; i < PPT; ++i) {
            const int p = t + i * NT;
            if (p < N && (dist[i] > best)) best = dist[i], arg = p;  // ascending p within the thread: first maximum
        }
#pragma unroll
        for (int ofs = 32; ofs > 0; ofs >>= 1) {
            const double ob = __shfl_xor(best, ofs);
            const int oa = __shfl_xor(arg, ofs);
            if (ob > best || (ob == best && oa < arg)) best = ob, arg = oa;
        }
        if (lane == 0) s_wmax[wave] = best, s_warg[wave] = arg;
        __syncthreads();
        best = s_wmax[0], arg = s_warg[0];
#pragma unroll
        for (int w = 1; w < 16; ++w) {
            const double ob = s_wmax[w];
            const int oa = s_warg[w];
            if (ob > best || (ob == best && oa < arg)) best = ob, arg = oa;
        }
        if (t == 0) s_sel[j] = arg;
        const double sx = (double)P[pt_off(layout, N, arg, 0)], sy = (double)P[pt_off(layout, N, arg, 1)],
                     sz = (double)P[pt_off(layout, N, arg, 2)];
#pragma unroll
        for (int i = 0; i < PPT; ++i) {
            const double dx = sx - px[i], dy = sy - py[i], dz = sz - pz[i];
            const double d = (dx * dx + dy * dy) + dz * dz;
            dist[i] = d < dist[i] ? d : dist[i];
        }
        __syncthreads();  // s_wmax / s_warg are rewritten by the next step
    }
    __syncthreads();
    for (int j = t; j < K; j += NT) {
        const int p = s_sel[j];
#pragma unroll
        for (int c = 0; c < 3; ++c) o[j * 3 + c] = P[pt_off(layout, N, p, c)];
    }
}

// xyz: (B,N,3) for SN_LAYOUT_BNC or (B,3,N) for SN_LAYOUT_BCN; idx (B,k) int32 in [0,N); out (B,k,3) fp32.
extern "C" int sn_nn_matching(int B, int N, int k, const float *xyz, int layout, const int *idx, int complete_fps, float *out,
                              sn_stream_t stream)
{
    SN_REQUIRE(B >= 0 && N >= 1 && k >= 1, "bad size");
    if (B == 0) return 0;
    SN_REQUIRE(xyz && idx && out, "null pointer");
    SN_REQUIRE(layout == SN_LAYOUT_BNC || layout == SN_LAYOUT_BCN, "bad layout");
    if (k > 1024 || N > 8192) return sn_set_error(SN_ERR_UNSUPPORTED, "sn_nn_matching: k <= 1024 and N <= 8192");
    hipStream_t st = (hipStream_t)stream;
#define SN_NM(PPT_) hipLaunchKernelGGL(nn_matching_kernel<PPT_>, dim3(B), dim3(1024), 0, st, N, k, layout, complete_fps, xyz, idx, out)
    if (N <= 1024) SN_NM(1);
    else if (N <= 2048) SN_NM(2);
    else if (N <= 4096) SN_NM(4);
    else SN_NM(8);
#undef SN_NM
    SN_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// Progressive sampler: nearest QUERY of every point for every nested prefix Q[:s_j] of the simplified cloud,
// from ONE pass over the M x N distances -- a thread owns a point, walks the queries in index order with a running
// (minimum, first argmin) and emits it whenever a prefix ends.  Same expression / strict-< tie rule as the Chamfer scan
// (chamfer_distance.cpp:59-112), so dist / idx of prefix j equal sn_chamfer_forward(Q[:, :s_j], P)'s dist2 / idx2 bit for bit.
// ------------------------------------------------------------------------------------------------
struct PrefixEnds {
    int n;
    int end[kMaxPrefixes];  // ascending, end[n-1] == M
};

__global__ void __launch_bounds__(256) prefix_colmin_kernel(int N, int M, const float *__restrict__ P, const float *__restrict__ Q,
                                                            PrefixEnds pe, float *__restrict__ dist, int *__restrict__ idx,
                                                            size_t stride)
{
    extern __shared__ float q_lds[];  // [M][3]
    const int b = blockIdx.y;
    for (int i = threadIdx.x; i < M * 3; i += blockDim.x) q_lds[i] = Q[(size_t)b * M * 3 + i];
    __syncthreads();
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const sn_xyz3 pv = *reinterpret_cast<const sn_xyz3 *>(P + ((size_t)b * N + n) * 3);
    float best = 0.f;
    int besti = 0, j = 0;
    for (int m = 0; m < M; ++m) {
        const float dx = q_lds[m * 3 + 0] - pv.x, dy = q_lds[m * 3 + 1] - pv.y, dz = q_lds[m * 3 + 2] - pv.z;
        const float d = (dx * dx + dy * dy) + dz * dz;
        if (m == 0 || d < best) best = d, besti = m;
        if (m + 1 == pe.end[j]) {
            dist[j * stride + (size_t)b * N + n] = best;
            idx[j * stride + (size_t)b * N + n] = besti;
            ++j;
        }
    }
}

// The same products with EIGHT threads per point (kPcSeg query ranges of ceil(M / 8) queries, one per thread): prefix_colmin_kernel has one
// thread per point walking all M queries -- at B = 32, N = 1024 that is 512 waves on 1024 SIMDs for 256 dependent steps each (28.6 us
// in BASELINE configs[4]'s step).  A thread leaves, per prefix, the running minimum of ITS range as it stood at the prefix's end (the
// whole range's when the prefix ends behind it, nothing when it ends in front); the ranges are then joined in ascending order with
// the strict compare of the one-thread walk -- minimum and arg-minimum are associative, the lowest query still wins a tie: same bits.
constexpr int kPcSeg = 8, kPcPts = 32;  // 256 threads: 32 points x 8 ranges
__global__ void __launch_bounds__(256) prefix_colmin_seg_kernel(int N, int M, const float *__restrict__ P, const float *__restrict__ Q,
                                                                PrefixEnds pe, float *__restrict__ dist, int *__restrict__ idx,
                                                                size_t stride)
{
    extern __shared__ float q_lds[];  // [M][3], then the snapshots
    float *sd = q_lds + (size_t)M * 3;                                         // [kPcPts][kMaxPrefixes][kPcSeg] distances
    int *si = reinterpret_cast<int *>(sd + kPcPts * kMaxPrefixes * kPcSeg);   // ... and queries
    const int b = blockIdx.y;
    for (int i = threadIdx.x; i < M * 3; i += blockDim.x) q_lds[i] = Q[(size_t)b * M * 3 + i];
    __syncthreads();
    const int pl = threadIdx.x >> 3, k = threadIdx.x & 7;
    const int n = blockIdx.x * kPcPts + pl;
    const int L = (M + kPcSeg - 1) / kPcSeg, m0 = k * L, m1 = min(M, m0 + L);
    const int nc = n < N ? n : N - 1;
    const sn_xyz3 pv = *reinterpret_cast<const sn_xyz3 *>(P + ((size_t)b * N + nc) * 3);
    float best = INFINITY;
    int besti = 0, j = 0;
    auto snap = [&](int jj) { sd[(pl * kMaxPrefixes + jj) * kPcSeg + k] = best, si[(pl * kMaxPrefixes + jj) * kPcSeg + k] = besti; };
    while (j < pe.n && pe.end[j] <= m0) snap(j), ++j;  // prefixes that end in front of this range: nothing (+inf)
    for (int m = m0; m < m1; ++m) {
        const float dx = q_lds[m * 3 + 0] - pv.x, dy = q_lds[m * 3 + 1] - pv.y, dz = q_lds[m * 3 + 2] - pv.z;
        const float d = (dx * dx + dy * dy) + dz * dz;
        if (m == m0 || d < best) best = d, besti = m;
        if (j < pe.n && m + 1 == pe.end[j]) snap(j), ++j;
    }
    while (j < pe.n) snap(j), ++j;  // prefixes that end behind this range: all of it
    __syncthreads();
    if (n >= N) return;
    for (int jj = k; jj < pe.n; jj += kPcSeg) {  // thread k of a point joins the ranges of prefixes k, k + 8
        const float *d8 = sd + (pl * kMaxPrefixes + jj) * kPcSeg;
        const int *i8 = si + (pl * kMaxPrefixes + jj) * kPcSeg;
        float bd = d8[0];
        int bi = i8[0];
#pragma unroll
        for (int q = 1; q < kPcSeg; ++q)
            if (d8[q] < bd) bd = d8[q], bi = i8[q];
        dist[jj * stride + (size_t)b * N + n] = bd;
        idx[jj * stride + (size_t)b * N + n] = bi;
    }
}

extern "C" int sn_prefix_point_minima(int B, int N, int M, int nprefix, const int *prefix_sizes, const float *P, const float *Q,
                                      float *dist, int *idx, sn_stream_t stream)
{
    SN_REQUIRE(B >= 1 && N >= 1 && M >= 1 && nprefix >= 1 && nprefix <= kMaxPrefixes, "bad size");
    SN_REQUIRE(prefix_sizes && P && Q && dist && idx, "null pointer");
    SN_REQUIRE((size_t)M * 12 <= 64 * 1024, "at most 5461 queries (LDS-staged)");
    PrefixEnds pe{};
    pe.n = nprefix;
    for (int j = 0; j < nprefix; ++j) {
        pe.end[j] = prefix_sizes[j];
        SN_REQUIRE(pe.end[j] >= 1 && (j == 0 || pe.end[j] > pe.end[j - 1]), "prefix sizes must be ascending");
    }
    SN_REQUIRE(pe.end[nprefix - 1] == M, "the last prefix is the whole simplified cloud");
    // few clouds: eight threads per point (the chip is not full otherwise); many: one thread per point walks all the queries
    const size_t lds_seg = (size_t)M * 12 + (size_t)kPcPts * kMaxPrefixes * kPcSeg * 8;
    if ((long long)B * N < 256 * 1024 && M >= 16 && lds_seg <= 64 * 1024) {
        hipLaunchKernelGGL(prefix_colmin_seg_kernel, dim3((N + kPcPts - 1) / kPcPts, B), dim3(256), lds_seg, (hipStream_t)stream, N, M, P, Q,
                           pe, dist, idx, (size_t)B * N);
    } else {
        const dim3 grid((N + 255) / 256, B), block(256);
        hipLaunchKernelGGL(prefix_colmin_kernel, grid, block, (size_t)M * 12, (hipStream_t)stream, N, M, P, Q, pe, dist, idx,
                           (size_t)B * N);
    }
    SN_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// The progressive sampler's prefixes (classification/train_samplenet_progressive.py:157-234: every nested size s uses the first s
// points of the ONE simplified / projected set) as contiguous tensors, all of them in one launch, and the gradient of that --
// grad[b, m, :] = sum over the prefixes that contain point m of their gradients, ascending prefix order -- in one launch.  As
// torch ops this is a strided copy per prefix forward and zeros + copy + accumulate per prefix backward: 33 launches of the 177 in
// BASELINE configs[4]'s step (profiles/r06/config5_graph_timeline.txt).  src (B, M, C) -> dst[j] (B, size_j, C); element size 4 bytes.
// ------------------------------------------------------------------------------------------------
struct PrefixPtrs {
    void *p[kMaxPrefixes];
    int size[kMaxPrefixes];
    int n;
};
__global__ void __launch_bounds__(256) prefix_pack_kernel(int M, int C, const unsigned *__restrict__ src, PrefixPtrs d)
{
    const int b = blockIdx.y;
    const int tot = M * C;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += gridDim.x * blockDim.x) {
        const unsigned v = src[(size_t)b * tot + e];
        const int m = e / C;
        for (int j = 0; j < d.n; ++j)
            if (m < d.size[j] && d.p[j]) reinterpret_cast<unsigned *>(d.p[j])[(size_t)b * d.size[j] * C + e] = v;
    }
}
__global__ void __launch_bounds__(256) prefix_scatter_sum_kernel(int M, int C, PrefixPtrs g, float *__restrict__ out)
{
    const int b = blockIdx.y;
    const int tot = M * C;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += gridDim.x * blockDim.x) {
        const int m = e / C;
        float acc = 0.f;
        for (int j = 0; j < g.n; ++j)  // ascending prefix order; a prefix nobody differentiated contributes nothing
            if (m < g.size[j] && g.p[j]) acc += reinterpret_cast<const float *>(g.p[j])[(size_t)b * g.size[j] * C + e];
        out[(size_t)b * tot + e] = acc;
    }
}

// dst: HOST array of nprefix device pointers (entries may be NULL: that prefix is not wanted), each (B, sizes[j], C) of 4-byte elements
extern "C" int sn_prefix_pack(int B, int M, int C, int nprefix, const int *sizes, const void *src, void *const *dst, sn_stream_t stream)
{
    SN_REQUIRE(B >= 1 && M >= 1 && C >= 1 && nprefix >= 1 && nprefix <= kMaxPrefixes && sizes && src && dst, "bad argument");
    PrefixPtrs d{};
    d.n = nprefix;
    for (int j = 0; j < nprefix; ++j) {
        SN_REQUIRE(sizes[j] >= 1 && sizes[j] <= M, "prefix size outside [1, M]");
        d.p[j] = dst[j], d.size[j] = sizes[j];
    }
    const int blocks = std::min((M * C + 255) / 256, 64);
    hipLaunchKernelGGL(prefix_pack_kernel, dim3(blocks, B), dim3(256), 0, (hipStream_t)stream, M, C, (const unsigned *)src, d);
    SN_LAUNCH_CHECK();
    return 0;
}

// grads: HOST array of nprefix device pointers (NULL: no gradient came through that prefix), each (B, sizes[j], C) fp32 -> out (B, M, C)
extern "C" int sn_prefix_scatter_sum(int B, int M, int C, int nprefix, const int *sizes, const float *const *grads, float *out,
                                     sn_stream_t stream)
{
    SN_REQUIRE(B >= 1 && M >= 1 && C >= 1 && nprefix >= 1 && nprefix <= kMaxPrefixes && sizes && grads && out, "bad argument");
    PrefixPtrs g{};
    g.n = nprefix;
    for (int j = 0; j < nprefix; ++j) {
        SN_REQUIRE(sizes[j] >= 1 && sizes[j] <= M, "prefix size outside [1, M]");
        g.p[j] = const_cast<float *>(grads[j]), g.size[j] = sizes[j];
    }
    const int blocks = std::min((M * C + 255) / 256, 64);
    hipLaunchKernelGGL(prefix_scatter_sum_kernel, dim3(blocks, B), dim3(256), 0, (hipStream_t)stream, M, C, g, out);
    SN_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// Clouds of different sizes as ONE batch of equal-size clouds for a network whose only reduction over the points is a maximum (PCRNet's
// PointNetFeatures, registration/models/pcrnet.py:23-46: per-point layers without BatchNorm + max-pool): cloud j (B, s_j, C) is
// repeated cyclically up to `len` points -- out[(j B + b), m, :] = src_j[b, m mod s_j, :] -- which leaves every cloud's pooled features
// unchanged, bit for bit (the copies add values the maximum already contains; the first occurrence wins ties, so the pooling backward
// hands its gradient to the ORIGINAL row).  The progressive sampler's four prefixes (32 .. 256 points) then take one extractor pass of
// 128 clouds x 256 points instead of four latency-bound passes.  Backward: the copies' gradients added onto their originals, in order.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) cyclic_pad_cat_kernel(int B, int len, int C, PrefixPtrs s, float *__restrict__ out)
{
    const int jb = blockIdx.y, j = jb / B, b = jb - j * B;
    const int sz = s.size[j];
    const float *src = reinterpret_cast<const float *>(s.p[j]) + (size_t)b * sz * C;
    float *dst = out + (size_t)jb * len * C;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < len * C; e += gridDim.x * blockDim.x) {
        const int m = e / C, c = e - m * C;
        dst[e] = src[(m % sz) * C + c];
    }
}
__global__ void __launch_bounds__(256) cyclic_pad_cat_bwd_kernel(int B, int len, int C, const float *__restrict__ gout, PrefixPtrs g)
{
    const int jb = blockIdx.y, j = jb / B, b = jb - j * B;
    if (!g.p[j]) return;
    const int sz = g.size[j];
    float *dst = reinterpret_cast<float *>(g.p[j]) + (size_t)b * sz * C;
    const float *src = gout + (size_t)jb * len * C;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < sz * C; e += gridDim.x * blockDim.x) {
        const int m = e / C, c = e - m * C;
        float acc = 0.f;
        for (int r = m; r < len; r += sz) acc += src[r * C + c];  // the original first, then its copies in order
        dst[e] = acc;
    }
}

// src: HOST array of nclouds device pointers, cloud j (B, sizes[j], C) fp32 -> out (nclouds * B, len, C), len >= every size
extern "C" int sn_cyclic_pad_cat(int B, int len, int C, int nclouds, const int *sizes, const float *const *src, float *out,
                                 sn_stream_t stream)
{
    SN_REQUIRE(B >= 1 && len >= 1 && C >= 1 && nclouds >= 1 && nclouds <= kMaxPrefixes && sizes && src && out, "bad argument");
    PrefixPtrs s{};
    s.n = nclouds;
    for (int j = 0; j < nclouds; ++j) {
        SN_REQUIRE(sizes[j] >= 1 && sizes[j] <= len && src[j], "cloud size outside [1, len]");
        s.p[j] = const_cast<float *>(src[j]), s.size[j] = sizes[j];
    }
    const int blocks = std::min((len * C + 255) / 256, 16);
    hipLaunchKernelGGL(cyclic_pad_cat_kernel, dim3(blocks, nclouds * B), dim3(256), 0, (hipStream_t)stream, B, len, C, s, out);
    SN_LAUNCH_CHECK();
    return 0;
}
// grads: HOST array of nclouds device pointers (NULL: that cloud wants no gradient), cloud j (B, sizes[j], C), overwritten
extern "C" int sn_cyclic_pad_cat_backward(int B, int len, int C, int nclouds, const int *sizes, const float *grad_out,
                                          float *const *grads, sn_stream_t stream)
{
    SN_REQUIRE(B >= 1 && len >= 1 && C >= 1 && nclouds >= 1 && nclouds <= kMaxPrefixes && sizes && grad_out && grads, "bad argument");
    PrefixPtrs g{};
    g.n = nclouds;
    for (int j = 0; j < nclouds; ++j) {
        SN_REQUIRE(sizes[j] >= 1 && sizes[j] <= len, "cloud size outside [1, len]");
        g.p[j] = grads[j], g.size[j] = sizes[j];
    }
    const int blocks = std::min((len * C + 255) / 256, 16);
    hipLaunchKernelGGL(cyclic_pad_cat_bwd_kernel, dim3(blocks, nclouds * B), dim3(256), 0, (hipStream_t)stream, B, len, C, grad_out, g);
    SN_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// The simplification losses of the first P nested prefixes (samplenet.py:171-181 per prefix, summed ascending as classification/
// train_samplenet_progressive.py:204-216 adds them) behind ONE autograd node: forward = the per-cloud reductions of every prefix in
// one launch + one combining workgroup; backward = the gradient on the simplified cloud of all P terms in one launch.  Every prefix's
// term is reduced in simp_loss_partial_kernel's order and its gradient formed in chamfer_bwd_reg_kernel's order (own term first,
// then the reference points that chose the target, ascending); the terms are added ascending, their gradients largest prefix first
// (the order autograd accumulates the separate terms' gradients in): the same bits as P separate sn_simplification_loss_forward /
// _backward calls on contiguous copies of the prefixes, without the copies (12 launches -> 3 at BASELINE configs[4]).  dq / iq (B,M): the per-query products of the FULL simplified set (they do not depend on the prefix);
// d2 / i2 (S,B,N): sn_prefix_point_minima's products, prefix p at p * B * N.
// ------------------------------------------------------------------------------------------------
struct PrefixLoss {
    int n;
    int size[kMaxPrefixes];
    float w[kMaxPrefixes];   // gamma + delta * size
    float c1[kMaxPrefixes];  // 1 / (B size), formed on the host like sn_simplification_loss_backward's
    float c2[kMaxPrefixes];  // w / (B N)
    float cm;                // 1 / B
};

__global__ void __launch_bounds__(256) prefix_simp_partial_kernel(int M, int N, PrefixLoss pl, const float *__restrict__ dq,
                                                                  const float *__restrict__ d2, float *__restrict__ part,
                                                                  int *__restrict__ argmax1)
{
    __shared__ float r0[256], r1[256], r2[256];
    __shared__ int ri[256];
    const int b = blockIdx.x, p = blockIdx.y, B = gridDim.x, t = threadIdx.x;
    const int n1 = pl.size[p];
    const float *d1 = dq + (size_t)b * M, *dd = d2 + ((size_t)p * B + b) * N;
    float s1 = 0.f, mx = -INFINITY, s2 = 0.f;
    int am = 0;
    for (int j = t; j < n1; j += 256) {
        const float v = d1[j];
        s1 += v;
        if (v > mx) mx = v, am = j;
    }
    for (int j = t; j < N; j += 256) s2 += dd[j];
    r0[t] = s1, r1[t] = mx, r2[t] = s2, ri[t] = am;
    for (int s = 128; s > 0; s >>= 1) {
        __syncthreads();
        if (t < s) {
            r0[t] += r0[t + s];
            r2[t] += r2[t + s];
            if (r1[t + s] > r1[t] || (r1[t + s] == r1[t] && ri[t + s] < ri[t])) r1[t] = r1[t + s], ri[t] = ri[t + s];
        }
    }
    if (t == 0) {
        float *o = part + ((size_t)p * B + b) * 3;
        o[0] = r0[0], o[1] = r1[0], o[2] = r2[0];
        argmax1[p * B + b] = ri[0];
    }
}

__global__ void __launch_bounds__(64) prefix_simp_final_kernel(int B, int N, PrefixLoss pl, const float *__restrict__ part,
                                                               float *__restrict__ loss)
{
    __shared__ float term[kMaxPrefixes];
    const int p = threadIdx.x;
    if (p < pl.n) {
        const float *q = part + (size_t)p * B * 3;
        float s1 = 0.f, mx = 0.f, s2 = 0.f;
        for (int b = 0; b < B; ++b) s1 += q[b * 3], mx += q[b * 3 + 1], s2 += q[b * 3 + 2];
        const float c12 = s1 / ((float)B * (float)pl.size[p]), cmax = mx / (float)B, c21 = s2 / ((float)B * (float)N);
        term[p] = c12 + cmax + pl.w[p] * c21;
    }
    __syncthreads();
    if (p == 0) {
        float tot = term[0];
        for (int j = 1; j < pl.n; ++j) tot += term[j];
        loss[0] = tot;
    }
}

template <int PPL>
__global__ void __launch_bounds__(256) prefix_simp_bwd_kernel(int M, int N, PrefixLoss pl, const float *__restrict__ T,
                                                              const float *__restrict__ S, const int *__restrict__ iq,
                                                              const int *__restrict__ i2, const int *__restrict__ argmax1,
                                                              const float *__restrict__ gL, float *__restrict__ gradT)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwaves = blockDim.x >> 6;
    const int b = blockIdx.x, B = gridDim.x;
    T += (size_t)b * M * 3, S += (size_t)b * N * 3, iq += (size_t)b * M, gradT += (size_t)b * M * 3;
    const float gLv = *gL;
    float sx[PPL], sy[PPL], sz[PPL];
#pragma unroll
    for (int i = 0; i < PPL; ++i) {
        const int l = i * 64 + lane;
        const sn_xyz3 sv = *reinterpret_cast<const sn_xyz3 *>(S + (size_t)(l < N ? l : 0) * 3);
        sx[i] = sv.x, sy[i] = sv.y, sz[i] = sv.z;
    }
    for (int j = blockIdx.y * nwaves + wave; j < M; j += gridDim.y * nwaves) {
        const float tx = T[j * 3], ty = T[j * 3 + 1], tz = T[j * 3 + 2];
        const int j2 = iq[j];
        const float dx = tx - S[j2 * 3 + 0], dy = ty - S[j2 * 3 + 1], dz = tz - S[j2 * 3 + 2];
        float gx = 0.f, gy = 0.f, gz = 0.f;
        for (int p = pl.n - 1; p >= 0; --p) {  // largest prefix first: the order autograd adds the terms' gradients in (last term first)
            const int n1 = pl.size[p];
            if (j >= n1) continue;  // (wave-uniform)
            const float g = gLv * (pl.c1[p] + (j == argmax1[p * B + b] ? pl.cm : 0.f)) * 2;
            const float gg = gLv * (pl.c2[p] + 0.f) * 2;
            float ax = g * dx, ay = g * dy, az = g * dz;
            const int *is = i2 + ((size_t)p * B + b) * N;
#pragma unroll
            for (int i = 0; i < PPL; ++i) {
                const int l = i * 64 + lane;
                sn_u64 mask = __ballot(l < N && is[l < N ? l : 0] == j);
                if (mask) {
                    const float cx = gg * (sx[i] - tx), cy = gg * (sy[i] - ty), cz = gg * (sz[i] - tz);
                    while (mask) {
                        const int t = __builtin_ctzll(mask);
                        mask &= mask - 1;
                        ax -= readlane_f(cx, t);
                        ay -= readlane_f(cy, t);
                        az -= readlane_f(cz, t);
                    }
                }
            }
            gx += ax, gy += ay, gz += az;
        }
        if (lane < 3) gradT[j * 3 + lane] = lane == 0 ? gx : (lane == 1 ? gy : gz);
    }
}

static int prefix_loss_args(int B, int M, int N, int nterms, const int *sizes, const float *weights, PrefixLoss &pl)
{
    SN_REQUIRE(nterms >= 1 && nterms <= kMaxPrefixes && sizes && weights, "1..16 prefix terms");
    pl.n = nterms;
    for (int j = 0; j < nterms; ++j) {
        SN_REQUIRE(sizes[j] >= 1 && sizes[j] <= M, "prefix size outside [1, M]");
        pl.size[j] = sizes[j], pl.w[j] = weights[j];
        pl.c1[j] = 1.0f / ((float)B * (float)sizes[j]), pl.c2[j] = weights[j] / ((float)B * (float)N);
    }
    pl.cm = 1.0f / (float)B;
    return 0;
}

// partial: 3 * nterms * B floats, argmax1: nterms * B ints (kept for the backward), loss: device scalar = sum over the nterms
// first prefixes of [mean(dq[:, :s]) + mean_b(max dq[:, :s]) + w * mean(d2[p])]
extern "C" int sn_prefix_simplification_loss_forward(int B, int M, int N, int nterms, const int *sizes, const float *weights,
                                                     const float *dq, const float *d2, float *partial, int *argmax1, float *loss,
                                                     sn_stream_t stream)
{
    SN_REQUIRE(B >= 1 && M >= 1 && N >= 1 && dq && d2 && partial && argmax1 && loss, "bad argument");
    PrefixLoss pl{};
    if (int rc = prefix_loss_args(B, M, N, nterms, sizes, weights, pl)) return rc;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(prefix_simp_partial_kernel, dim3(B, nterms), dim3(256), 0, st, M, N, pl, dq, d2, partial, argmax1);
    hipLaunchKernelGGL(prefix_simp_final_kernel, dim3(1), dim3(64), 0, st, B, N, pl, partial, loss);
    SN_LAUNCH_CHECK();
    return 0;
}

// grad_samp (B,M,3) = d loss / d samp_pc * *grad_loss (rows >= the largest of the nterms sizes: 0); N <= 2048
extern "C" int sn_prefix_simplification_loss_backward(int B, int M, int N, int nterms, const int *sizes, const float *weights,
                                                      const float *samp_pc, const float *ref_pc, const int *iq, const int *i2,
                                                      const int *argmax1, const float *grad_loss, float *grad_samp, sn_stream_t stream)
{
    SN_REQUIRE(B >= 1 && M >= 1 && N >= 1 && N <= 2048, "bad size (N <= 2048)");
    SN_REQUIRE(samp_pc && ref_pc && iq && i2 && argmax1 && grad_loss && grad_samp, "null pointer");
    PrefixLoss pl{};
    if (int rc = prefix_loss_args(B, M, N, nterms, sizes, weights, pl)) return rc;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(B, std::max(1, std::min((M + 3) / 4, (kChamferBwdGroups + B - 1) / B))), block(256);
#define SN_PB(PPL_) \
    hipLaunchKernelGGL(prefix_simp_bwd_kernel<PPL_>, grid, block, 0, st, M, N, pl, samp_pc, ref_pc, iq, i2, argmax1, grad_loss, grad_samp)
    if (N <= 64) SN_PB(1);
    else if (N <= 256) SN_PB(4);
    else if (N <= 1024) SN_PB(16);
    else SN_PB(32);
#undef SN_PB
    SN_LAUNCH_CHECK();
    return 0;
}

#ifdef SN_CS_TIMELINE
// copies the first nblocks x 16 stamps of the last chamfer_soft_bwd_kernel launch to the host and clears them
extern "C" int sn_debug_chamfer_soft_bwd_timeline(void *host, int nblocks)
{
    if (nblocks < 0 || nblocks > 8192) return -1;
    if (hipDeviceSynchronize() != hipSuccess) return -2;
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(g_cs_tl), (size_t)nblocks * 16 * 8) != hipSuccess) return -3;
    static unsigned long long zeros[8192 * 16];
    return hipMemcpyToSymbol(HIP_SYMBOL(g_cs_tl), zeros, sizeof(zeros)) == hipSuccess ? 0 : -4;
}
#endif
