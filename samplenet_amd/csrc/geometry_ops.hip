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

#include "sn_common.h"

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
__global__ void __launch_bounds__(256) chamfer_bwd_kernel(int nt, int ns, const float *__restrict__ T,
                                                          const float *__restrict__ S,
                                                          const float *__restrict__ gT,
                                                          const int *__restrict__ idxT,
                                                          const float *__restrict__ gS,
                                                          const int *__restrict__ idxS, float *__restrict__ gradT,
                                                          int own_first)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwaves = blockDim.x >> 6;
    const int b = blockIdx.x;
    T += (size_t)b * nt * 3, S += (size_t)b * ns * 3;
    gT += (size_t)b * nt, idxT += (size_t)b * nt, gS += (size_t)b * ns, idxS += (size_t)b * ns;
    gradT += (size_t)b * nt * 3;
    for (int j = blockIdx.y * nwaves + wave; j < nt; j += gridDim.y * nwaves) {
        const float tx = T[j * 3 + 0], ty = T[j * 3 + 1], tz = T[j * 3 + 2];
        const int j2 = idxT[j];
        const float g = gT[j] * 2;
        const float ox = g * (tx - S[j2 * 3 + 0]);
        const float oy = g * (ty - S[j2 * 3 + 1]);
        const float oz = g * (tz - S[j2 * 3 + 2]);
        float ax = 0.f, ay = 0.f, az = 0.f;
        if (own_first) ax += ox, ay += oy, az += oz;
        for (int l0 = 0; l0 < ns; l0 += 64) {
            const int l = l0 + lane;
            sn_u64 mask = __ballot(l < ns && idxS[l] == j);
            while (mask) {
                const int ll = l0 + __builtin_ctzll(mask);
                mask &= mask - 1;
                const float gg = gS[ll] * 2;
                ax -= gg * (S[ll * 3 + 0] - tx);
                ay -= gg * (S[ll * 3 + 1] - ty);
                az -= gg * (S[ll * 3 + 2] - tz);
            }
        }
        if (!own_first) ax += ox, ay += oy, az += oz;
        if (lane < 3) gradT[j * 3 + lane] = lane == 0 ? ax : (lane == 1 ? ay : az);
    }
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
    float *grad_sigma_partial;
};

template <bool FUSED>
__global__ void __launch_bounds__(256) soft_bwd_kernel(SoftBwdArgs a)
{
    __shared__ float s_part[4];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwaves = blockDim.x >> 6;
    const int b = blockIdx.x;
    const int n = a.n, m = a.m, K = a.k;
    const float *__restrict__ Pb = a.P + (size_t)b * 3 * n;
    const float *__restrict__ Qb = a.Q + (size_t)b * 3 * m;
    const float T = *a.temperature;
    const float sigma = fmaxf(T * T, a.min_sigma);
    float gsig = 0.f;  // this wave's share of d loss / d sigma

    for (int j = wave; j < m; j += nwaves) {
        const float qx = Qb[pt_off(a.q_layout, m, j, 0)];
        const float qy = Qb[pt_off(a.q_layout, m, j, 1)];
        const float qz = Qb[pt_off(a.q_layout, m, j, 2)];
        const bool act = lane < K;
        const int id = act ? a.idx[((size_t)b * m + j) * K + lane] : 0;
        float gx = 0.f, gy = 0.f, gz = 0.f;
        if (act) {
            gx = Pb[pt_off(a.p_layout, n, id, 0)];
            gy = Pb[pt_off(a.p_layout, n, id, 1)];
            gz = Pb[pt_off(a.p_layout, n, id, 2)];
        }
        const float dx = gx - qx, dy = gy - qy, dz = gz - qz;
        const float d = (dx * dx + dy * dy) + dz * dz;
        const float s = act ? -(d / sigma) : -INFINITY;
        float mx = s;
#pragma unroll
        for (int t = 1; t < 64; t <<= 1) mx = fmaxf(mx, __shfl_xor(mx, t));
        const float e = act ? expf(s - mx) : 0.f;
        float den = 0.f;
        for (int t = 0; t < K; ++t) den += readlane_f(e, t);
        const float w = e / den;

        float gw;  // d loss / d w_t
        float go0 = 0.f, go1 = 0.f, go2 = 0.f;
        if (FUSED) {
            const float *gp = a.grad_proj + (size_t)b * 3 * m;
            go0 = gp[pt_off(a.gproj_layout, m, j, 0)];
            go1 = gp[pt_off(a.gproj_layout, m, j, 1)];
            go2 = gp[pt_off(a.gproj_layout, m, j, 2)];
            gw = (go0 * gx + go1 * gy) + go2 * gz;
        } else {
            gw = act ? a.grad_weights[((size_t)b * m + j) * K + lane] : 0.f;
        }
        float dot = 0.f;
        for (int t = 0; t < K; ++t) dot += readlane_f(w, t) * readlane_f(gw, t);
        const float gs = act ? w * (gw - dot) : 0.f;  // softmax backward
        const float gd = -gs / sigma;                 // s = -d / sigma
        const float cx = 2.0f * gd * dx, cy = 2.0f * gd * dy, cz = 2.0f * gd * dz;
        const float sg = act ? gs * d / (sigma * sigma) : 0.f;
        float aqx = 0.f, aqy = 0.f, aqz = 0.f, asg = 0.f;
        for (int t = 0; t < K; ++t) {
            aqx -= readlane_f(cx, t);
            aqy -= readlane_f(cy, t);
            aqz -= readlane_f(cz, t);
            asg += readlane_f(sg, t);
        }
        gsig += asg;
        if (a.grad_Q && lane < 3) {
            const float o = lane == 0 ? aqx : (lane == 1 ? aqy : aqz);
            a.grad_Q[(size_t)b * 3 * m + pt_off(a.gq_layout, m, j, lane)] = o;
        }
        if (a.grad_P && act) {
            float *gpb = a.grad_P + (size_t)b * 3 * n;
            const int lay = FUSED ? a.p_layout : SN_LAYOUT_BCN;
            const float ex = FUSED ? go0 * w : 0.f, ey = FUSED ? go1 * w : 0.f, ez = FUSED ? go2 * w : 0.f;
            atomicAdd(&gpb[pt_off(lay, n, id, 0)], ex + cx);
            atomicAdd(&gpb[pt_off(lay, n, id, 1)], ey + cy);
            atomicAdd(&gpb[pt_off(lay, n, id, 2)], ez + cz);
        }
    }
    // fixed-order block reduction of the sigma gradient: wave 0..3 in order
    if (lane == 0) s_part[wave] = gsig;
    __syncthreads();
    if (threadIdx.x == 0 && a.grad_sigma_partial) {
        float tot = 0.f;
        for (int w2 = 0; w2 < nwaves; ++w2) tot += s_part[w2];
        a.grad_sigma_partial[b] = tot;
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
    const float sigma = fmaxf(T * T, min_sigma);
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
                                                                  float *__restrict__ gw, float *__restrict__ gX)
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
        if (gX) atomicAdd(&gX[((size_t)b * c + ch) * n + id], g * wt);
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

__global__ void __launch_bounds__(256) group_point_grad_kernel(int n, int c, int m, int ns,
                                                               const float *__restrict__ grad_out,
                                                               const int *__restrict__ idx,
                                                               float *__restrict__ grad_points)
{
    const int b = blockIdx.y;
    const size_t tot = (size_t)m * ns * c;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += (size_t)gridDim.x * blockDim.x) {
        const int l = (int)(e % c);
        const size_t jk = e / c;
        const int ii = idx[(size_t)b * m * ns + jk];
        atomicAdd(&grad_points[((size_t)b * n + ii) * c + l], grad_out[(size_t)b * tot + e]);
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

__global__ void __launch_bounds__(256) grouping_operation_grad_kernel(int c, int n, int m, int ns,
                                                                      const float *__restrict__ grad_out,
                                                                      const int *__restrict__ idx,
                                                                      float *__restrict__ grad_feat)
{
    const int b = blockIdx.y;
    const size_t mk = (size_t)m * ns, tot = (size_t)c * mk;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += (size_t)gridDim.x * blockDim.x) {
        const int ch = (int)(e / mk);
        const size_t jk = e % mk;
        const int ii = idx[(size_t)b * mk + jk];
        atomicAdd(&grad_feat[((size_t)b * c + ch) * n + ii], grad_out[(size_t)b * tot + e]);
    }
}

static inline unsigned grid_for(size_t tot, int block = 256, unsigned cap = 4096)
{
    size_t g = (tot + block - 1) / block;
    return (unsigned)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace sn

using namespace sn;

extern "C" int sn_chamfer_backward(int b, int n, const float *xyz1, int m, const float *xyz2,
                                   const float *grad_dist1, const int *idx1, const float *grad_dist2,
                                   const int *idx2, float *grad_xyz1, float *grad_xyz2, sn_stream_t stream)
{
    SN_REQUIRE(b >= 0 && n >= 0 && m >= 0, "negative size");
    if (b == 0 || n == 0 || m == 0) return 0;
    SN_REQUIRE(xyz1 && xyz2 && grad_dist1 && grad_dist2 && idx1 && idx2, "null input");
    hipStream_t st = (hipStream_t)stream;
    auto ysplit = [&](int nt) { return std::max(1, std::min((nt + 3) / 4, (1024 + b - 1) / b)); };
    if (grad_xyz1)
        hipLaunchKernelGGL(chamfer_bwd_kernel, dim3(b, ysplit(n)), dim3(256), 0, st, n, m, xyz1, xyz2, grad_dist1,
                           idx1, grad_dist2, idx2, grad_xyz1, 1);
    if (grad_xyz2)
        hipLaunchKernelGGL(chamfer_bwd_kernel, dim3(b, ysplit(m)), dim3(256), 0, st, m, n, xyz2, xyz1, grad_dist2,
                           idx2, grad_dist1, idx1, grad_xyz2, 0);
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
    hipLaunchKernelGGL(soft_bwd_kernel<true>, dim3(b), dim3(256), 0, (hipStream_t)stream, a);
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
    hipLaunchKernelGGL(soft_bwd_kernel<false>, dim3(b), dim3(256), 0, (hipStream_t)stream, a);
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
    hipError_t e = hipMemsetAsync(grad_points, 0, sizeof(float) * (size_t)b * n * c, (hipStream_t)stream);
    if (e != hipSuccess) return sn_set_error((int)e, "sn_group_point_grad: %s", hipGetErrorString(e));
    const size_t tot = (size_t)m * nsample * c;
    if (tot == 0) return 0;
    SN_REQUIRE(grad_out && idx, "null pointer");
    hipLaunchKernelGGL(group_point_grad_kernel, dim3(grid_for(tot), b), dim3(256), 0, (hipStream_t)stream, n, c, m,
                       nsample, grad_out, idx, grad_points);
    SN_LAUNCH_CHECK();
    return 0;
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
    hipError_t e = hipMemsetAsync(grad_features, 0, sizeof(float) * (size_t)b * n * c, (hipStream_t)stream);
    if (e != hipSuccess) return sn_set_error((int)e, "sn_grouping_operation_grad: %s", hipGetErrorString(e));
    const size_t tot = (size_t)m * nsample * c;
    if (tot == 0) return 0;
    SN_REQUIRE(grad_out && idx, "null pointer");
    hipLaunchKernelGGL(grouping_operation_grad_kernel, dim3(grid_for(tot), b), dim3(256), 0, (hipStream_t)stream, c,
                       n, m, nsample, grad_out, idx, grad_features);
    SN_LAUNCH_CHECK();
    return 0;
}
