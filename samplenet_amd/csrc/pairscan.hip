// pairscan.hip -- the fused pair-scan kernel of the SampleNet hot path (gfx950 / CDNA4).
//
// One pass over the M x N squared-distance matrix of a cloud yields everything geometric the
// reference computes in three separate packages:
//   kNN top-K of every query          knn_cuda.KNN  (soft_projection.py:11-14; in-tree definition
//                                     tf_grouping.py:64-91 + tf_grouping_g.cu:83-123)
//   row min/argmin   (dist_q, idx_q)  ChamferDistanceKernel, 1st launch (chamfer_distance.cu:149)
//   column min/argmin (dist_p, idx_p) ChamferDistanceKernel, 2nd launch (chamfer_distance.cu:150)
//   soft projection of the query      soft_projection.py:138-152
//
// Mapping (wave64):  a wave owns one QUERY at a time (coordinates wave-uniform -> SGPR operands),
// its 64 lanes own the dataset POINTS, PPL points per lane held in registers for the whole block
// lifetime (N <= 64*PPL: "single chunk"), so the cloud is read from HBM/L2 exactly once per wave.
//   column minima : per-lane running (min, argmin) over the wave's queries  -> free of cross-lane work
//   top-K / row min: threshold filter.  tau = max over G lane-groups of the group's minimum
//                   (G = pow2 >= K) is an upper bound of the K-th smallest distance that costs
//                   6 cross-lane steps; the ~K..3K candidates with d <= tau are compacted to LDS
//                   as 64-bit (distance, index) keys and ranked by counting.  Keys are unique, so
//                   the order is exactly ascending (distance, index) and ties resolve to the lowest
//                   index, as the reference's strict '<' ascending scans do.
// Work per pair is ~8 VALU for the distance (never contracted: see sqdist) + ~3 for the column
// minimum + ~6 amortised for filter/compaction; nothing is re-read from memory.
//
// Multi-chunk mode (N > 64*PPL_MAX): the wave walks the cloud in chunks keeping its running top-K
// in LDS; column minima are then produced by a second launch with the roles swapped.
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "sn_common.h"

// The distance expression must round exactly like the reference's CPU code
// (chamfer_distance.cpp:74-77: float products and sums, no FMA).  hipcc contracts by default.
#pragma clang fp contract(off)

namespace sn {

constexpr int kListCap = 192;  // per-wave candidate list entries (3 per lane)
constexpr int kListPitch = kListCap + kWave;  // + one scrap entry per lane / overrun of the last round (branch-free compaction)

// Debug build only (-DSN_PS_TIMELINE, tools/pairscan_timeline.py): thread 0 of every workgroup stamps the 100 MHz wall clock at
// the phase boundaries of its first query.
#ifdef SN_PS_TIMELINE
__device__ unsigned long long g_ps_tl[8192 * 16];
#define PS_TL(slot)                                                                                              \
    do {                                                                                                         \
        if (threadIdx.x == 0) g_ps_tl[((blockIdx.y * gridDim.x + blockIdx.x) & 8191) * 16 + (slot)] = wall_clock64(); \
    } while (0)
#if SN_PS_TIMELINE >= 2  // serialise the phases: every stamp waits for the memory operations before it
#define PS_TL_DRAIN() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory")
#else
#define PS_TL_DRAIN()
#endif
#else
#define PS_TL(slot)
#define PS_TL_DRAIN()
#endif

__device__ __forceinline__ float sqdist(float qx, float qy, float qz, float px, float py, float pz)
{
    const float dx = px - qx, dy = py - qy, dz = pz - qz;
    const float xx = dx * dx, yy = dy * dy, zz = dz * dz;
    const float s = xx + yy;
    return s + zz;
}

// Keep the K smallest of list[0..cnt) in list[0..min(K,cnt)) sorted ascending (rank by counting;
// keys are unique).  Wave-synchronous: every lane of the wave calls it with the same arguments.
__device__ __forceinline__ int merge_topk(sn_u64 *list, int cnt, int K, int lane)
{
    constexpr int R = kListCap / kWave;
    sn_u64 mine[R];
    int rank[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int e = lane + r * kWave;
        mine[r] = e < cnt ? list[e] : kKeyInf;
        rank[r] = 0;
    }
    for (int t = 0; t < cnt; ++t) {
        const sn_u64 k = list[t];  // same address for all lanes: LDS broadcast
#pragma unroll
        for (int r = 0; r < R; ++r) rank[r] += (k < mine[r]) ? 1 : 0;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int e = lane + r * kWave;
        if (e < cnt && rank[r] < K) list[rank[r]] = mine[r];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    return cnt < K ? cnt : K;
}

// Load this lane's PPL points of the chunk starting at c0 (point i*64+lane); lanes past the end of
// the cloud get +inf coordinates, i.e. distance +inf: never selected.
struct sn_xyz {
    float x, y, z;
};

template <int PPL>
__device__ __forceinline__ void load_chunk(float (&px)[PPL], float (&py)[PPL], float (&pz)[PPL],
                                           const float *__restrict__ Pb, int layout, int N, int c0, int lane)
{
    if (layout == SN_LAYOUT_BNC) {
        // point-major cloud: one 12-byte load per point (global_load_dwordx3; consecutive lanes read consecutive points)
        // instead of three strided dword loads -- 16 instead of 48 memory instructions per lane for a 1024-point cloud
#pragma unroll
        for (int i = 0; i < PPL; ++i) {
            const int n = c0 + i * kWave + lane;
            const sn_xyz v = *reinterpret_cast<const sn_xyz *>(Pb + (size_t)(n < N ? n : 0) * 3);
            px[i] = v.x, py[i] = v.y, pz[i] = v.z;
        }
        asm volatile("" ::: "memory");  // all PPL loads in flight before the first select waits for one (in every build: the
                                        // debug build with the phase stamps otherwise paired each load with its select)
#pragma unroll
        for (int i = 0; i < PPL; ++i) {
            const bool ok = c0 + i * kWave + lane < N;
            px[i] = ok ? px[i] : INFINITY;
            py[i] = ok ? py[i] : INFINITY;
            pz[i] = ok ? pz[i] : INFINITY;
        }
        return;
    }
    // channel-major cloud: three coalesced dword loads per point, all 3 * PPL of them in flight before the first use
    const float *__restrict__ Px = Pb, *__restrict__ Py = Pb + N, *__restrict__ Pz = Pb + 2 * (size_t)N;
#pragma unroll
    for (int i = 0; i < PPL; ++i) {
        const int n = c0 + i * kWave + lane;
        const int nn = n < N ? n : 0;
        px[i] = Px[nn];
        py[i] = Py[nn];
        pz[i] = Pz[nn];
    }
    asm volatile("" ::: "memory");  // (keeps the selects below from being paired with their loads, one wait per point)
#pragma unroll
    for (int i = 0; i < PPL; ++i) {
        const bool ok = c0 + i * kWave + lane < N;
        px[i] = ok ? px[i] : INFINITY;
        py[i] = ok ? py[i] : INFINITY;
        pz[i] = ok ? pz[i] : INFINITY;
    }
}

// Workgroup size cap per variant: registers hold 3*PPL coordinates + PPL distances (+ 2*PPL column
// minima), so the wider variants trade waves per SIMD for VGPRs (no spills at any size).
constexpr int max_threads(int ppl, bool colmin)
{
    return (ppl >= 32 && colmin) ? 256 : ((ppl >= 32 || (ppl == 16 && colmin)) ? 512 : 1024);
}

// ---- cross-lane helpers on the DPP path (no LDS round trip) --------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ float dpp_f(float x)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(x), __float_as_int(x), CTRL, 0xf, 0xf, false));
}
constexpr int kDppXor1 = 0xB1, kDppXor2 = 0x4E, kDppHalfMirror = 0x141, kDppRowMirror = 0x140;
// value of the lane below within the row of 16 (lane 0 of a row reads 0): one step of a sequential scan
__device__ __forceinline__ float dpp_shr1(float x)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x111, 0xf, 0xf, true));
}

// Threshold tau = max over the 2^LOGG lane groups (contiguous groups of 64 >> LOGG lanes) of the group minimum of x.
// x >= +0 (or +inf), so the cross-row step can run on the SALU as unsigned integer min/max of the bit patterns.
template <int LOGG>
__device__ __forceinline__ float group_minmax(float x)
{
    // within a row of 16 lanes: xor1, xor2 (quads), half-mirror (8), mirror (16)
    if (LOGG <= 5) x = fminf(x, dpp_f<kDppXor1>(x)); else x = fmaxf(x, dpp_f<kDppXor1>(x));          // group >= 2 lanes
    if (LOGG <= 4) x = fminf(x, dpp_f<kDppXor2>(x)); else x = fmaxf(x, dpp_f<kDppXor2>(x));          // group >= 4
    if (LOGG <= 3) x = fminf(x, dpp_f<kDppHalfMirror>(x)); else x = fmaxf(x, dpp_f<kDppHalfMirror>(x));  // group >= 8
    if (LOGG <= 2) x = fminf(x, dpp_f<kDppRowMirror>(x)); else x = fmaxf(x, dpp_f<kDppRowMirror>(x));    // group >= 16
    const unsigned r0 = __builtin_amdgcn_readlane(__float_as_int(x), 0), r1 = __builtin_amdgcn_readlane(__float_as_int(x), 16);
    const unsigned r2 = __builtin_amdgcn_readlane(__float_as_int(x), 32), r3 = __builtin_amdgcn_readlane(__float_as_int(x), 48);
    unsigned a, b;
    if (LOGG <= 1) a = r0 < r1 ? r0 : r1, b = r2 < r3 ? r2 : r3; else a = r0 > r1 ? r0 : r1, b = r2 > r3 ? r2 : r3;  // group >= 32
    const unsigned t = (LOGG == 0) ? (a < b ? a : b) : (a > b ? a : b);                                              // group == 64
    return __uint_as_float(t);
}

// Rank the wave's cnt <= 64 candidates list[0..cnt) (entries list[cnt .. cnt+15] hold kKeyInf) and move them to lane ==
// rank: afterwards lane t < min(cnt, 64) holds the t-th smallest key.  Keys are unique, so ranks are a permutation.  Every
// lane reads the same entry (LDS broadcast) and counts the keys below its own: two VALU per candidate.
__device__ __forceinline__ sn_u64 rank_to_lanes(const sn_u64 *list, int cnt, int lane)
{
    const sn_u64 mine = (lane < cnt) ? list[lane] : kKeyInf;
    int rank = 0;
    for (int u = 0; u < cnt; u += 16) {
#pragma unroll
        for (int v = 0; v < 16; ++v) rank += (list[u + v] < mine) ? 1 : 0;
    }
    const int dst = (lane < cnt) ? rank : lane;  // lanes without a candidate keep to themselves (no collision)
    const unsigned plo = (unsigned)__builtin_amdgcn_ds_permute(dst << 2, (int)(unsigned)mine);
    const unsigned phi = (unsigned)__builtin_amdgcn_ds_permute(dst << 2, (int)(unsigned)(mine >> 32));
    return ((sn_u64)phi << 32) | plo;
}

struct PairscanArgs {
    const float *P;
    const float *Q;
    int p_layout, q_layout;
    int B, N, M, K;  // K >= 1 internally (K == 1 when only the nearest neighbour is wanted)
    int *knn_idx;
    float *knn_dist;
    float *dist_q;
    int *idx_q;
    float *dist_p;
    int *idx_p;
    float *proj;
    int proj_layout;
    float *weights;
    const float *temperature;
    float min_sigma;
    sn_u64 *colmin_ws;  // [B][gridDim.y][N] partial column minima as (distance, query) keys when the queries of a
                        // cloud are spread over several workgroups (gridDim.y > 1); finalised by colmin_finalize_kernel
    // Optional: the queries are not read but PRODUCED here, as the last fully connected layer of the sampler's head
    // (samplenet.py:103-104: y = fc4(relu(bn_fc3(z3))), y.view(B, 3, M)): query j of cloud b has coordinate c =
    // bias[c*M + j] + sum_k relu(z3[b][k] * scale[k] + shift[k]) * W[c*M + j][k].  The wave that scans query j computes it
    // (3 rows x fc_k MACs spread over the lanes) and stores it to q_out in q_layout.
    const float *fc_z, *fc_scale, *fc_shift, *fc_w, *fc_bias;
    int fc_k;
    float *q_out;
    // Optional (gridDim.y > 1): the per-point minima are combined ACROSS the workgroups of a cloud by 64-bit atomicMax on
    // INVERTED keys (max of ~key = min of key; max / min are associative and commutative, so the result does not depend on
    // the arrival order) into colmin_keys [B][N], which the caller keeps zeroed between steps -- no partial key sets, no
    // finalisation launch.  Each workgroup then also leaves its share of the step loss's query-side reductions:
    // qpart [B][gridDim.y][2] = (sum dist_q, sum proj) over its queries, qmax [B][gridDim.y] = max of (dist_q, ~query) keys.
    sn_u64 *colmin_keys;
    float *qpart;
    sn_u64 *qmax;
    // Optional: only the first q_valid[b / q_group] queries of cloud b are scanned (the rest are cyclic copies of them,
    // sn_cyclic_pad_cat: a copy never wins a per-point minimum, and its own products are read by nobody) -- dist_q / idx_q / the kNN
    // outputs of the others are left unwritten.
    const int *q_valid;
    int q_group;
};

// coordinate c of query j (see PairscanArgs::fc_w): lane partial sums in k order, then a fixed xor tree
__device__ __forceinline__ void fc_query(const PairscanArgs &a, int b, int j, int lane, float &qx, float &qy, float &qz)
{
    const int Kf = a.fc_k, M = a.M;
    const float *z = a.fc_z + (size_t)b * Kf;
    const float *w0 = a.fc_w + (size_t)j * Kf, *w1 = w0 + (size_t)M * Kf, *w2 = w1 + (size_t)M * Kf;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    for (int k = lane * 4; k < Kf; k += kWave * 4) {
        const float4 zv = *reinterpret_cast<const float4 *>(z + k);
        const float4 sc = *reinterpret_cast<const float4 *>(a.fc_scale + k);
        const float4 sh = *reinterpret_cast<const float4 *>(a.fc_shift + k);
        const float4 u = *reinterpret_cast<const float4 *>(w0 + k);
        const float4 v = *reinterpret_cast<const float4 *>(w1 + k);
        const float4 w = *reinterpret_cast<const float4 *>(w2 + k);
        // (u < 0 ? 0 : u, not fmaxf: a NaN from diverged BatchNorm statistics must reach the loss, as in the MLP kernels)
        const float ux = fmaf(zv.x, sc.x, sh.x), uy = fmaf(zv.y, sc.y, sh.y), uz = fmaf(zv.z, sc.z, sh.z), uw = fmaf(zv.w, sc.w, sh.w);
        const float ax = ux < 0.f ? 0.f : ux, ay = uy < 0.f ? 0.f : uy, az = uz < 0.f ? 0.f : uz, aw = uw < 0.f ? 0.f : uw;
        s0 = fmaf(aw, u.w, fmaf(az, u.z, fmaf(ay, u.y, fmaf(ax, u.x, s0))));
        s1 = fmaf(aw, v.w, fmaf(az, v.z, fmaf(ay, v.y, fmaf(ax, v.x, s1))));
        s2 = fmaf(aw, w.w, fmaf(az, w.z, fmaf(ay, w.y, fmaf(ax, w.x, s2))));
    }
    // xor tree 32, 16, 8, 4, 2, 1.  The last four levels stay inside a row of 16 lanes and run on the DPP path: after the
    // level before, lanes l and l ^ 8 (then l ^ 4) hold the same value, so a rotation by 8 (by 4) hands every lane the very
    // partner value of the xor butterfly -- same sums, same bits, no LDS round trip.
#pragma unroll
    for (int o = 32; o >= 16; o >>= 1) {
        s0 += __shfl_xor(s0, o);
        s1 += __shfl_xor(s1, o);
        s2 += __shfl_xor(s2, o);
    }
    s0 += dpp_f<0x128>(s0), s1 += dpp_f<0x128>(s1), s2 += dpp_f<0x128>(s2);                    // row_ror:8
    s0 += dpp_f<0x124>(s0), s1 += dpp_f<0x124>(s1), s2 += dpp_f<0x124>(s2);                    // row_ror:4
    s0 += dpp_f<kDppXor2>(s0), s1 += dpp_f<kDppXor2>(s1), s2 += dpp_f<kDppXor2>(s2);           // quad_perm [2,3,0,1]
    s0 += dpp_f<kDppXor1>(s0), s1 += dpp_f<kDppXor1>(s1), s2 += dpp_f<kDppXor1>(s2);           // quad_perm [1,0,3,2]
    qx = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, s0 + a.fc_bias[j])));
    qy = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, s1 + a.fc_bias[M + j])));
    qz = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, s2 + a.fc_bias[2 * M + j])));
    if (lane == 0) {
        float *qo = a.q_out + (size_t)b * 3 * M;
        qo[pt_off(a.q_layout, M, j, 0)] = qx;
        qo[pt_off(a.q_layout, M, j, 1)] = qy;
        qo[pt_off(a.q_layout, M, j, 2)] = qz;
    }
}

template <int PPL, bool SINGLE, bool COLMIN, int LOGG>
__global__ void __launch_bounds__(max_threads(PPL, COLMIN)) pairscan_kernel(PairscanArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwaves = blockDim.x >> 6;
    const int b = blockIdx.x;
    const int N = a.N, M = a.M, K = a.K;
    const int Mv = a.q_valid ? a.q_valid[b / a.q_group] : M;  // (queries to scan: the valid ones, spread over the cloud's workgroups)
    const int qpb = (Mv + gridDim.y - 1) / gridDim.y;
    const int q0 = blockIdx.y * qpb;
    const int q1 = min(Mv, q0 + qpb);
    PS_TL(0);

    sn_u64 *list = reinterpret_cast<sn_u64 *>(smem) + wave * kListPitch;
    sn_u64 *colmin = reinterpret_cast<sn_u64 *>(smem) + nwaves * kListPitch;  // [64*PPL] when COLMIN

    const float *__restrict__ Pb = a.P + (size_t)b * 3 * N;
    const float *__restrict__ Qb = a.Q + (size_t)b * 3 * M;

    const bool want_soft = (a.proj != nullptr) || (a.weights != nullptr);

    // (no barrier here: colmin[] is next touched behind the query loop, and the barrier in front of that use orders these
    //  writes -- a barrier in the prologue would hold the cloud loads back until the slowest wave's arguments arrived)
    if (COLMIN) {
        for (int i = threadIdx.x; i < kWave * PPL; i += blockDim.x) colmin[i] = kKeyInf;
    }

    float px[PPL], py[PPL], pz[PPL];
    float cd[PPL];  // column minima (COLMIN)
    int ci[PPL];
#pragma unroll
    for (int i = 0; i < PPL; ++i) {
        cd[i] = INFINITY;
        ci[i] = 0;
    }
    if (SINGLE) load_chunk<PPL>(px, py, pz, Pb, a.p_layout, N, 0, lane);
    PS_TL_DRAIN();
    PS_TL(1);

    float wsum_dq = 0.f, wsum_pj = 0.f;  // colmin_keys mode: this wave's share of sum dist_q, sum proj, max (dist_q, ~query)
    sn_u64 wmax = 0;
    for (int j = q0 + wave; j < q1; j += nwaves) {  // wave-uniform
        float qx, qy, qz;
        if (a.fc_w) {
            fc_query(a, b, j, lane, qx, qy, qz);
        } else {
            qx = Qb[pt_off(a.q_layout, M, j, 0)];
            qy = Qb[pt_off(a.q_layout, M, j, 1)];
            qz = Qb[pt_off(a.q_layout, M, j, 2)];
        }
        if (j == q0) { PS_TL_DRAIN(); PS_TL(2); }
        int cnt = 0;
        float thr_run = INFINITY;
        float temp_q = 1.0f;

        for (int c0 = 0; c0 < N; c0 += kWave * PPL) {
            if (!SINGLE) load_chunk<PPL>(px, py, pz, Pb, a.p_layout, N, c0, lane);
            float d[PPL];
            float lmin = INFINITY;
#pragma unroll
            for (int i = 0; i < PPL; ++i) {
                d[i] = sqdist(qx, qy, qz, px[i], py[i], pz[i]);
                lmin = fminf(lmin, d[i]);
                if (COLMIN) {
                    if (d[i] < cd[i]) {  // strict '<', ascending j: lowest query index wins
                        cd[i] = d[i];
                        ci[i] = j;
                    }
                }
            }
            // the temperature, for the softmax at the end of the query: read here, behind the cloud's and the query's
            // loads and through a vector load (a scalar load in the prologue would hold all of them back until it lands)
            if (want_soft && c0 == 0) {
                int zero = 0;
                asm volatile("" : "+v"(zero));
                temp_q = a.temperature[zero];
            }
            // tau = max over the 2^LOGG lane groups of the group's minimum (2^LOGG >= K): >= K points lie at or below it
            float thr = fminf(group_minmax<LOGG>(lmin), thr_run);
            if (j == q0) PS_TL(3);
            // compact candidates (d <= thr) into the wave's LDS list, round by round with a capacity check (a merge
            // when the list would overflow): the general form
            auto compact_checked = [&]() {
#pragma unroll
                for (int i = 0; i < PPL; ++i) {
                    const int n = c0 + i * kWave + lane;
                    const bool pred = (n < N) && (d[i] <= thr);
                    const sn_u64 mask = __builtin_amdgcn_ballot_w64(pred);
                    if (mask != 0) {
                        if (cnt + kWave > kListCap) {
                            cnt = merge_topk(list, cnt, K, lane);
                            if (cnt == K) thr = fminf(thr, key_dist(list[K - 1]));
                        }
                        const int pos = cnt + __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32),
                                                                        __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
                        if (pred) list[pos] = make_key(d[i], n);
                        cnt += __builtin_popcountll(mask);
                    }
                }
            };
            if constexpr (SINGLE) {
                // whole cloud in registers, list empty: the same rounds without a branch -- positions from a running
                // count on the SALU, lanes without a candidate write a scrap entry behind the list.  The ~K..3K
                // candidates of a real cloud fit the list; a round that starts within the list may run up to 64 entries
                // past its end (the scrap entries' space), and the total tells (clouds of coincident points), in which
                // case the checked form redoes the compaction.
                auto compact_all = [&](auto full) {
                    int base = 0;
                    int lane_q = lane;  // re-materialised per query: keeps the PPL point indices out of long-lived registers
                    asm volatile("" : "+v"(lane_q));
#pragma unroll
                    for (int i = 0; i < PPL; ++i) {
                        const int n = i * kWave + lane_q;
                        const bool pred = decltype(full)::value ? (d[i] <= thr) : ((n < N) && (d[i] <= thr));
                        const sn_u64 mask = __builtin_amdgcn_ballot_w64(pred);
                        const int at = base < kListCap ? base : kListCap;  // clamped only when the total overflows anyway
                        const int pos = at + __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32),
                                                                       __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
                        // (spelling the select out as v_cndmask on the ballot lets the rounds overlap -- compaction 0.92 ->
                        //  0.64 us at B = 32 -- but costs 22 VGPRs: 2 instead of 3 waves per SIMD, 15 -> 12 M clouds/s
                        //  when the batch fills the chip)
                        list[pred ? pos : kListCap + lane] = make_key(d[i], n);  // others: the lane's own scrap entry
                        base += __builtin_popcountll(mask);
                    }
                    return base;
                };
                // (lanes past the end of a cloud hold +inf distances; they pass d <= thr only when thr itself is +inf,
                //  so the bound check can go when every lane's every point exists)
                const int total = (N == kWave * PPL) ? compact_all(std::true_type{}) : compact_all(std::false_type{});
                if (total <= kListCap) {
                    cnt = total;
                    if (lane < 16) list[total + lane] = kKeyInf;  // rank_to_lanes reads whole groups of 16
                } else {
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    compact_checked();
                    if (cnt <= kWave && lane < 16) list[cnt + lane] = kKeyInf;  // (as above: the merges may leave <= 64 entries)
                }
            } else {
                compact_checked();
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            if (j == q0) { PS_TL_DRAIN(); PS_TL(4); }
            if (!SINGLE) {
                cnt = merge_topk(list, cnt, K, lane);
                if (cnt == K) thr_run = key_dist(list[K - 1]);
            }
        }
        // ---- per-query outputs: lanes t < K hold neighbour t (ascending (distance, index)) ----
        sn_u64 key;
        if (SINGLE && cnt <= kWave) {
            // usual case (~K..3K candidates): one candidate per lane, ranked by counting, moved to lane == rank
            key = rank_to_lanes(list, cnt, lane);
            cnt = cnt < K ? cnt : K;
            if (lane >= cnt) key = kKeyInf;
        } else {
            if (SINGLE) cnt = merge_topk(list, cnt, K, lane);
            key = (lane < cnt) ? list[lane] : kKeyInf;
        }
        const int nidx = (lane < cnt) ? key_index(key) : 0;
        const float nd = (lane < cnt) ? key_dist(key) : INFINITY;
        if (j == q0) PS_TL(5);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        const size_t qrow = (size_t)b * M + j;
        if (lane < K) {
            if (a.knn_idx) a.knn_idx[qrow * K + lane] = nidx;
            if (a.knn_dist) a.knn_dist[qrow * K + lane] = nd;
        }
        if (lane == 0) {
            if (a.dist_q) a.dist_q[qrow] = nd;
            if (a.idx_q) a.idx_q[qrow] = nidx;
            wsum_dq += nd;
            const sn_u64 kq = make_key(nd, (int)(0xFFFFFFFFu - (unsigned)j));  // larger distance first, then lower query
            wmax = kq > wmax ? kq : wmax;
        }
        if (want_soft) {
            const float sigma = sn_sigma(temp_q, a.min_sigma);   // soft_projection.py:97-99
            const float s = (lane < cnt) ? -(nd / sigma) : -INFINITY;  // soft_projection.py:92-95
            const float mx = readlane_f(s, 0);  // neighbours ascend in distance: lane 0 holds the maximum of s
            const float e = (lane < cnt) ? expf(s - mx) : 0.f;
            if constexpr (LOGG <= 4) {
                // K <= 16: neighbour t sits in lane t of row 0.  Rows 0..2 (16 lanes each) take one coordinate each, and
                // the ascending-k sums (denominator :143, weighted sum :148-151) run as sequential DPP scans inside the
                // rows -- lane t adds its term to the running sum of lane t-1, the very order of the reference's loop --
                // instead of K trips of v_readlane.  Steps past K-1 leave lane K-1 unchanged.
                constexpr int kSteps = (1 << LOGG) - 1;
                const int sub = lane & 15, row = lane >> 4;
                const int nidx_r = __builtin_amdgcn_ds_bpermute(sub << 2, nidx);
                float g = 0.f;
                if (a.proj && row < 3 && sub < cnt) g = Pb[pt_off(a.p_layout, N, nidx_r, row)];
                float den = e;
#pragma unroll
                for (int t = 0; t < kSteps; ++t) den = e + dpp_shr1(den);
                const float w = e / readlane_f(den, K - 1);  // softmax over the K neighbours (:143)
                if (a.weights && lane < K) a.weights[qrow * K + lane] = w;
                if (a.proj) {
                    const float term = g * __int_as_float(__builtin_amdgcn_ds_bpermute(sub << 2, __float_as_int(w)));
                    float o = term;  // product then sum, ascending k
#pragma unroll
                    for (int t = 0; t < kSteps; ++t) o = term + dpp_shr1(o);
                    if (row < 3 && sub == K - 1) a.proj[(size_t)b * 3 * M + pt_off(a.proj_layout, M, j, row)] = o;
                    if (a.qpart) {
                        const float ox = readlane_f(o, K - 1), oy = readlane_f(o, 16 + K - 1), oz = readlane_f(o, 32 + K - 1);
                        if (lane == 0) wsum_pj += (ox + oy) + oz;
                    }
                }
            } else {
                float gx = 0.f, gy = 0.f, gz = 0.f;
                if (lane < cnt) {
                    gx = Pb[pt_off(a.p_layout, N, nidx, 0)];
                    gy = Pb[pt_off(a.p_layout, N, nidx, 1)];
                    gz = Pb[pt_off(a.p_layout, N, nidx, 2)];
                }
                float den = 0.f;
                for (int t = 0; t < K; ++t) den += readlane_f(e, t);
                const float w = e / den;  // softmax over the K neighbours (:143)
                if (a.weights && lane < K) a.weights[qrow * K + lane] = w;
                if (a.proj) {
                    float ox = 0.f, oy = 0.f, oz = 0.f;
                    for (int t = 0; t < K; ++t) {  // ascending k, product then sum (:148-151)
                        const float wt = readlane_f(w, t);
                        ox += readlane_f(gx, t) * wt;
                        oy += readlane_f(gy, t) * wt;
                        oz += readlane_f(gz, t) * wt;
                    }
                    if (lane < 3) {
                        const float o = lane == 0 ? ox : (lane == 1 ? oy : oz);
                        a.proj[(size_t)b * 3 * M + pt_off(a.proj_layout, M, j, lane)] = o;
                    }
                    if (lane == 0) wsum_pj += (ox + oy) + oz;
                }
            }
        }
        if (j == q0) { PS_TL_DRAIN(); PS_TL(6); }
    }

    if (COLMIN) {
        // this wave's share of the query-side reductions (keys mode), then the waves' column minima: unsigned min of
        // (distance, query index) keys
        float *wq = reinterpret_cast<float *>(colmin + kWave * PPL);  // [nwaves][2]
        sn_u64 *wk = colmin + kWave * PPL + nwaves;                   // [nwaves]
        if (a.colmin_keys && lane == 0) wq[wave * 2] = wsum_dq, wq[wave * 2 + 1] = wsum_pj, wk[wave] = wmax;
        __syncthreads();  // colmin[] initialised by every wave
#pragma unroll
        for (int i = 0; i < PPL; ++i) atomicMin(&colmin[i * kWave + lane], make_key(cd[i], ci[i]));
        __syncthreads();
        PS_TL(7);
        if (gridDim.y == 1) {
            for (int n = threadIdx.x; n < N; n += blockDim.x) {
                const sn_u64 k = colmin[n];
                if (a.dist_p) a.dist_p[(size_t)b * N + n] = key_dist(k);
                if (a.idx_p) a.idx_p[(size_t)b * N + n] = key_index(k);
            }
        } else if (a.colmin_keys) {  // combine with the cloud's other workgroups in place (see PairscanArgs)
            sn_u64 *kk = a.colmin_keys + (size_t)b * N;
            for (int n = threadIdx.x; n < N; n += blockDim.x) atomicMax(kk + n, ~colmin[n]);
            PS_TL_DRAIN();
            PS_TL(8);
            // query-side reductions of this workgroup, waves in order
            if (wave == 0) {  // lane w holds wave w's partials (one LDS round trip), summed in wave order
                const bool has = lane < nwaves;
                const float pd = has ? wq[lane * 2] : 0.f, pp = has ? wq[lane * 2 + 1] : 0.f;
                const sn_u64 pk = has ? wk[lane] : 0;
                float sd = 0.f, sp = 0.f;
                sn_u64 mk = 0;
                for (int w2 = 0; w2 < nwaves; ++w2) {
                    sd += readlane_f(pd, w2), sp += readlane_f(pp, w2);
                    const sn_u64 kw = ((sn_u64)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(pk >> 32), w2) << 32) |
                                      (unsigned)__builtin_amdgcn_readlane((int)(unsigned)pk, w2);
                    mk = kw > mk ? kw : mk;
                }
                if (lane == 0) {
                    const size_t o = (size_t)b * gridDim.y + blockIdx.y;
                    a.qpart[o * 2] = sd, a.qpart[o * 2 + 1] = sp;
                    a.qmax[o] = mk;
                }
            }
        } else {  // this workgroup saw only its share of the queries: publish partial keys
            sn_u64 *ws = a.colmin_ws + ((size_t)b * gridDim.y + blockIdx.y) * N;
            for (int n = threadIdx.x; n < N; n += blockDim.x) ws[n] = colmin[n];
        }
    }
    PS_TL_DRAIN();
    PS_TL(9);
}

// dist_p / idx_p = minimum over the G partial keys of every point (min of (distance, query) keys = lowest query on ties)
__global__ void __launch_bounds__(256) colmin_finalize_kernel(int N, int G, const sn_u64 *__restrict__ ws,
                                                              float *__restrict__ dist_p, int *__restrict__ idx_p)
{
    const int b = blockIdx.y;
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    sn_u64 k = kKeyInf;
    for (int g = 0; g < G; ++g) {
        const sn_u64 v = ws[((size_t)b * G + g) * N + n];
        k = v < k ? v : k;
    }
    if (dist_p) dist_p[(size_t)b * N + n] = key_dist(k);
    if (idx_p) idx_p[(size_t)b * N + n] = key_index(k);
}

template <int PPL, bool SINGLE, bool COLMIN>
static int launch_pairscan(const PairscanArgs &a, int waves, int ysplit, hipStream_t st)
{
    const size_t lds = (size_t)waves * kListPitch * 8 + (COLMIN ? (size_t)kWave * PPL * 8 + (size_t)waves * 16 : 0);
    const dim3 grid(a.B, ysplit), block(waves * kWave);
    // number of lane groups for the threshold = power of two >= K (instantiated: 1, 8, 16, 64)
    if (a.K <= 1)
        hipLaunchKernelGGL((pairscan_kernel<PPL, SINGLE, COLMIN, 0>), grid, block, lds, st, a);
    else if (a.K <= 8)
        hipLaunchKernelGGL((pairscan_kernel<PPL, SINGLE, COLMIN, 3>), grid, block, lds, st, a);
    else if (a.K <= 16)
        hipLaunchKernelGGL((pairscan_kernel<PPL, SINGLE, COLMIN, 4>), grid, block, lds, st, a);
    else
        hipLaunchKernelGGL((pairscan_kernel<PPL, SINGLE, COLMIN, 6>), grid, block, lds, st, a);
    return 0;
}

// number of workgroups the queries of one cloud are spread over (single-chunk path, N <= 2048): the batch alone cannot
// fill the chip at B = 32, so aim at >= 2 workgroups per CU
int pairscan_ysplit(int B, int N, int M, bool colmin)
{
    if (N > kWave * 32) return 1;
    const int ppl = N <= 64 ? 1 : (N <= 256 ? 4 : (N <= 1024 ? 16 : 32));
    const int maxw = max_threads(ppl, colmin) / kWave;
    const int want = (512 + B - 1) / std::max(B, 1);
    return std::max(1, std::min(want, (M + maxw - 1) / maxw));
}

// Host-side dispatch.  colmin => dist_p / idx_p requested.  ws/ws_bytes: optional scratch that lets the queries of a
// cloud be spread over several workgroups even when column minima are wanted (partials + a finalize kernel).
int pairscan_dispatch(PairscanArgs a, void *ws, long long ws_bytes, bool finalize, int *used_split, hipStream_t st)
{
    const bool colmin = a.dist_p || a.idx_p || (ws && !finalize) || a.colmin_keys;
    const int N = a.N, M = a.M;
    if (used_split) *used_split = 1;
    if (N <= kWave * 32) {
        // single chunk: the whole cloud lives in the wave's registers
        const int ppl = N <= 64 ? 1 : (N <= 256 ? 4 : (N <= 1024 ? 16 : 32));
        const int maxw = max_threads(ppl, colmin) / kWave;
        int ysplit = pairscan_ysplit(a.B, N, M, colmin);
        if (colmin && ysplit > 1 && !a.colmin_keys) {
            const long long need = (long long)a.B * ysplit * N * 8;
            if (!ws || ws_bytes < need) ysplit = 1;
        }
        const int qpb = (M + ysplit - 1) / ysplit;
        int waves = std::max(1, std::min(maxw, qpb));  // one query per wave at the sampler's sizes (swept 2 / 4 / 8: 8 is fastest)
        // the batch alone fills the chip: workgroups of 4 waves pack a CU's register file better (the 1024-point variant
        // holds 153 VGPRs -> 3 waves per SIMD = 12 per CU: three workgroups of 4 instead of one of 8; swept 8 / 6 / 4 /
        // 3 / 2 at B = 8192: 10.1 / 8.4 / 15.0 / 14.4 / 14.3 M clouds/s)
        if (ysplit == 1 && a.B >= 512) waves = std::min(waves, 4);
        a.colmin_ws = (colmin && ysplit > 1 && !a.colmin_keys) ? (sn_u64 *)ws : nullptr;
        if (used_split) *used_split = ysplit;
#define SN_PS(PPL_)                                                                      \
    (colmin ? launch_pairscan<PPL_, true, true>(a, waves, ysplit, st)                    \
            : launch_pairscan<PPL_, true, false>(a, waves, ysplit, st))
        switch (ppl) {
            case 1: SN_PS(1); break;
            case 4: SN_PS(4); break;
            case 16: SN_PS(16); break;
            default: SN_PS(32); break;
        }
#undef SN_PS
        if (colmin && ysplit > 1 && finalize && !a.colmin_keys)
            hipLaunchKernelGGL(colmin_finalize_kernel, dim3((N + 255) / 256, a.B), dim3(256), 0, st, N, ysplit,
                               (const sn_u64 *)ws, a.dist_p, a.idx_p);
        return 0;
    }
    // multi chunk: row products only; the caller obtains column minima by a swapped second call
    if (colmin) return sn_set_error(SN_ERR_UNSUPPORTED, "pairscan: column minima need N <= 2048 (internal)");
    const int want = (1024 + a.B - 1) / a.B;
    const int ysplit = std::max(1, std::min(want, (M + 15) / 16));
    const int qpb = (M + ysplit - 1) / ysplit;
    const int waves = std::max(1, std::min(16, qpb));
    return launch_pairscan<16, false, false>(a, waves, ysplit, st);
}

}  // namespace sn

using sn::PairscanArgs;

extern "C" long long sn_pairscan_workspace_bytes(int B, int N, int M)
{
    if (N > sn::kWave * 32) return 0;
    const int ysplit = std::max(1, std::min((512 + B - 1) / std::max(B, 1), M));
    return ysplit > 1 ? (long long)B * ysplit * N * 8 : 0;
}

extern "C" int sn_pairscan_forward_ws(int B, int N, int M, int K, const float *P, int p_layout, const float *Q,
                                      int q_layout, int *knn_idx, float *knn_dist, float *dist_q, int *idx_q,
                                      float *dist_p, int *idx_p, float *proj, int proj_layout, float *weights,
                                      const float *temperature, float min_sigma, void *workspace,
                                      long long workspace_bytes, sn_stream_t stream);

extern "C" int sn_pairscan_forward(int B, int N, int M, int K, const float *P, int p_layout, const float *Q,
                                   int q_layout, int *knn_idx, float *knn_dist, float *dist_q, int *idx_q,
                                   float *dist_p, int *idx_p, float *proj, int proj_layout, float *weights,
                                   const float *temperature, float min_sigma, sn_stream_t stream)
{
    return sn_pairscan_forward_ws(B, N, M, K, P, p_layout, Q, q_layout, knn_idx, knn_dist, dist_q, idx_q, dist_p, idx_p,
                                  proj, proj_layout, weights, temperature, min_sigma, nullptr, 0, stream);
}

extern "C" int sn_pairscan_forward_ws(int B, int N, int M, int K, const float *P, int p_layout, const float *Q,
                                      int q_layout, int *knn_idx, float *knn_dist, float *dist_q, int *idx_q,
                                      float *dist_p, int *idx_p, float *proj, int proj_layout, float *weights,
                                      const float *temperature, float min_sigma, void *workspace,
                                      long long workspace_bytes, sn_stream_t stream)
{
    SN_REQUIRE(B >= 0 && N >= 0 && M >= 0 && K >= 0, "negative size");
    if (B == 0 || (M == 0 && N == 0)) return 0;
    SN_REQUIRE(P && Q, "null point cloud");
    SN_REQUIRE(N >= 1 && M >= 1, "empty cloud on one side only");
    SN_REQUIRE(K <= N, "K exceeds the number of dataset points");
    SN_REQUIRE(K <= 64, "K > 64 unsupported");
    SN_REQUIRE(p_layout == SN_LAYOUT_BNC || p_layout == SN_LAYOUT_BCN, "bad p_layout");
    SN_REQUIRE(q_layout == SN_LAYOUT_BNC || q_layout == SN_LAYOUT_BCN, "bad q_layout");
    const bool knn = knn_idx || knn_dist, soft = proj || weights;
    SN_REQUIRE(!(knn || soft) || K >= 1, "K must be >= 1 for kNN / projection outputs");
    SN_REQUIRE(!soft || temperature, "projection needs the temperature pointer");
    hipStream_t st = (hipStream_t)stream;
    PairscanArgs a{};
    a.P = P, a.Q = Q, a.p_layout = p_layout, a.q_layout = q_layout;
    a.B = B, a.N = N, a.M = M, a.K = K < 1 ? 1 : K;
    a.knn_idx = knn_idx, a.knn_dist = knn_dist, a.dist_q = dist_q, a.idx_q = idx_q;
    a.proj = proj, a.proj_layout = proj_layout, a.weights = weights;
    a.temperature = temperature, a.min_sigma = min_sigma;
    const bool colmin = dist_p || idx_p;
    if (N <= sn::kWave * 32 || !colmin) {
        a.dist_p = dist_p, a.idx_p = idx_p;
        int rc = sn::pairscan_dispatch(a, workspace, workspace_bytes, true, nullptr, st);
        if (rc) return rc;
    } else {
        int rc = sn::pairscan_dispatch(a, nullptr, 0, true, nullptr, st);  // row products
        if (rc) return rc;
        PairscanArgs s{};  // column minima = row minima of the swapped problem
        s.P = Q, s.Q = P, s.p_layout = q_layout, s.q_layout = p_layout;
        s.B = B, s.N = M, s.M = N, s.K = 1;
        s.dist_q = dist_p, s.idx_q = idx_p;
        rc = sn::pairscan_dispatch(s, nullptr, 0, true, nullptr, st);
        if (rc) return rc;
    }
    SN_LAUNCH_CHECK();
    return 0;
}

extern "C" int sn_chamfer_forward(int b, int n, const float *xyz, int m, const float *xyz2, float *result,
                                  int *result_i, float *result2, int *result2_i, sn_stream_t stream)
{
    SN_REQUIRE(b >= 0 && n >= 0 && m >= 0, "negative size");
    if (b == 0 || (n == 0 && m == 0)) return 0;
    SN_REQUIRE(n >= 1 && m >= 1, "empty cloud on one side only");
    // lanes own the larger set (it amortises the per-query filter cost); outputs map accordingly
    if (m >= n)
        return sn_pairscan_forward(b, m, n, 0, xyz2, SN_LAYOUT_BNC, xyz, SN_LAYOUT_BNC, nullptr, nullptr, result,
                                   result_i, result2, result2_i, nullptr, 0, nullptr, nullptr, 0.f, stream);
    return sn_pairscan_forward(b, n, m, 0, xyz, SN_LAYOUT_BNC, xyz2, SN_LAYOUT_BNC, nullptr, nullptr, result2,
                               result2_i, result, result_i, nullptr, 0, nullptr, nullptr, 0.f, stream);
}

extern "C" int sn_knn(int b, int n, int m, int k, const float *xyz1, int layout1, const float *xyz2, int layout2,
                      int *idx, float *dist, sn_stream_t stream)
{
    SN_REQUIRE(k >= 1, "k must be >= 1");
    return sn_pairscan_forward(b, n, m, k, xyz1, layout1, xyz2, layout2, idx, dist, nullptr, nullptr, nullptr,
                               nullptr, nullptr, 0, nullptr, nullptr, 0.f, stream);
}

// Pair scan whose per-point minima are left as PARTIAL (distance,index) keys, one set per workgroup of a cloud:
// workspace [B][G][N] u64 with G = sn_pairscan_colmin_splits(B, N, M) (> 1, else use sn_pairscan_forward).  The consumer
// (sn_sampler_step_loss_forward) takes the minimum over G while it reduces the loss: one launch less than finalising here.
extern "C" int sn_pairscan_colmin_splits(int B, int N, int M)
{
    if (B < 1 || N < 1 || M < 1) return 0;
    return sn::pairscan_ysplit(B, N, M, true);
}

extern "C" int sn_pairscan_forward_partial(int B, int N, int M, int K, const float *P, int p_layout, const float *Q,
                                           int q_layout, int *knn_idx, float *dist_q, int *idx_q, float *proj,
                                           int proj_layout, const float *temperature, float min_sigma, void *workspace,
                                           long long workspace_bytes, sn_stream_t stream)
{
    SN_REQUIRE(B >= 1 && N >= 1 && M >= 1 && K >= 1 && K <= 64 && K <= N, "bad size");
    SN_REQUIRE(P && Q && workspace && temperature, "null pointer");
    SN_REQUIRE(p_layout == SN_LAYOUT_BNC || p_layout == SN_LAYOUT_BCN, "bad p_layout");
    SN_REQUIRE(q_layout == SN_LAYOUT_BNC || q_layout == SN_LAYOUT_BCN, "bad q_layout");
    const int G = sn_pairscan_colmin_splits(B, N, M);
    if (G <= 1 || N > sn::kWave * 32)
        return sn_set_error(SN_ERR_UNSUPPORTED, "sn_pairscan_forward_partial: this shape runs as one workgroup per cloud");
    SN_REQUIRE(workspace_bytes >= (long long)B * G * N * 8, "workspace too small");
    PairscanArgs a{};
    a.P = P, a.Q = Q, a.p_layout = p_layout, a.q_layout = q_layout;
    a.B = B, a.N = N, a.M = M, a.K = K;
    a.knn_idx = knn_idx, a.dist_q = dist_q, a.idx_q = idx_q;
    a.proj = proj, a.proj_layout = proj_layout, a.temperature = temperature, a.min_sigma = min_sigma;
    int used = 0;
    int rc = sn::pairscan_dispatch(a, workspace, workspace_bytes, false, &used, (hipStream_t)stream);
    if (rc) return rc;
    SN_LAUNCH_CHECK();  // (first: SN_REQUIRE discards the pending launch status)
    if (used != G) return sn_set_error(SN_ERR_BAD_ARGUMENT, "%s: internal: split mismatch", __func__);
    return 0;
}

// sn_chamfer_forward on a batch whose SMALLER clouds are padded with cyclic copies (sn_cyclic_pad_cat): only the first
// q_valid[b / q_group] points of the smaller cloud of pair b are scanned as queries (device array, one int per group of q_group clouds);
// dist / idx of the copies are left unwritten, every other product equals sn_chamfer_forward's.  Needs the clouds' roles fixed by the
// caller: xyz_small (B, m, 3) the padded side, xyz_large (B, n, 3), m <= n <= 2048.  workspace as sn_pairscan_workspace_bytes(B, n, m).
extern "C" int sn_chamfer_forward_valid(int B, int m, const float *xyz_small, int n, const float *xyz_large, const int *q_valid, int q_group,
                                        float *dist_small, int *idx_small, float *dist_large, int *idx_large, void *workspace,
                                        long long workspace_bytes, sn_stream_t stream)
{
    SN_REQUIRE(B >= 1 && m >= 1 && n >= m && n <= sn::kWave * 32 && q_group >= 1, "bad size (m <= n <= 2048)");
    SN_REQUIRE(xyz_small && xyz_large && q_valid && dist_small && idx_small && dist_large && idx_large, "null pointer");
    PairscanArgs a{};
    a.P = xyz_large, a.Q = xyz_small, a.p_layout = SN_LAYOUT_BNC, a.q_layout = SN_LAYOUT_BNC;
    a.B = B, a.N = n, a.M = m, a.K = 1;
    a.dist_q = dist_small, a.idx_q = idx_small, a.dist_p = dist_large, a.idx_p = idx_large;
    a.q_valid = q_valid, a.q_group = q_group;
    int rc = sn::pairscan_dispatch(a, workspace, workspace_bytes, true, nullptr, (hipStream_t)stream);
    if (rc) return rc;
    SN_LAUNCH_CHECK();
    return 0;
}

// Pair scan for callers that run the sampler step's backward right behind it (engine): per-point minima combined across a
// cloud's workgroups by atomicMax on inverted keys in colmin_keys [B][N] (u64, ZERO on entry; sn_sampler_step_loss_keys
// leaves it zero again), query-side loss partials in qpart [B][G][2] / qmax [B][G], G = sn_pairscan_colmin_splits(B,N,M) > 1.
// Queries: Q (B,3,M), or -- fc_w != NULL -- produced by the head's last layer as in sn_pairscan_forward_partial_fc (Q then
// receives them).  One launch; the loss side needs no reduction launch of its own.
extern "C" int sn_pairscan_forward_keys(int B, int N, int M, int K, const float *P, int p_layout, float *Q, const float *fc_z,
                                        const float *fc_scale, const float *fc_shift, const float *fc_w, const float *fc_bias,
                                        int Kfc, int *knn_idx, float *dist_q, int *idx_q, float *proj, int proj_layout,
                                        const float *temperature, float min_sigma, void *colmin_keys, float *qpart, void *qmax,
                                        sn_stream_t stream)
{
    SN_REQUIRE(B >= 1 && N >= 1 && M >= 1 && K >= 1 && K <= 64 && K <= N, "bad size");
    SN_REQUIRE(P && Q && dist_q && proj && temperature && colmin_keys && qpart && qmax, "null pointer");
    SN_REQUIRE(!fc_w || (fc_z && fc_scale && fc_shift && fc_bias && Kfc >= 4 && Kfc % 4 == 0), "bad fc operands");
    SN_REQUIRE(p_layout == SN_LAYOUT_BNC || p_layout == SN_LAYOUT_BCN, "bad p_layout");
    const int G = sn_pairscan_colmin_splits(B, N, M);
    if (G <= 1 || N > sn::kWave * 32)
        return sn_set_error(SN_ERR_UNSUPPORTED, "sn_pairscan_forward_keys: this shape runs as one workgroup per cloud");
    PairscanArgs a{};
    a.P = P, a.Q = Q, a.p_layout = p_layout, a.q_layout = SN_LAYOUT_BCN;
    a.B = B, a.N = N, a.M = M, a.K = K;
    a.knn_idx = knn_idx, a.dist_q = dist_q, a.idx_q = idx_q;
    a.proj = proj, a.proj_layout = proj_layout, a.temperature = temperature, a.min_sigma = min_sigma;
    if (fc_w) {
        a.fc_z = fc_z, a.fc_scale = fc_scale, a.fc_shift = fc_shift, a.fc_w = fc_w, a.fc_bias = fc_bias, a.fc_k = Kfc;
        a.q_out = Q;
    }
    a.colmin_keys = (sn_u64 *)colmin_keys, a.qpart = qpart, a.qmax = (sn_u64 *)qmax;
    int used = 0;
    int rc = sn::pairscan_dispatch(a, nullptr, 0, false, &used, (hipStream_t)stream);
    if (rc) return rc;
    SN_LAUNCH_CHECK();  // (first: SN_REQUIRE discards the pending launch status)
    if (used != G) return sn_set_error(SN_ERR_BAD_ARGUMENT, "%s: internal: split mismatch", __func__);
    return 0;
}

// sn_pairscan_forward_partial whose queries are produced in the scan itself by the head's last fully connected layer
// (PairscanArgs::fc_w): z3 (B, Kfc) pre-BatchNorm output of the layer before, scale/shift (Kfc) its BatchNorm as an affine
// map, W (3M, Kfc), bias (3M).  q_out (B,3,M) receives the simplified cloud.  One launch less than FC layer + scan.
extern "C" int sn_pairscan_forward_partial_fc(int B, int N, int M, int K, const float *P, int p_layout, const float *fc_z,
                                              const float *fc_scale, const float *fc_shift, const float *fc_w,
                                              const float *fc_bias, int Kfc, float *q_out, int *knn_idx, float *dist_q,
                                              int *idx_q, float *proj, int proj_layout, const float *temperature,
                                              float min_sigma, void *workspace, long long workspace_bytes,
                                              sn_stream_t stream)
{
    SN_REQUIRE(B >= 1 && N >= 1 && M >= 1 && K >= 1 && K <= 64 && K <= N, "bad size");
    SN_REQUIRE(P && fc_z && fc_scale && fc_shift && fc_w && fc_bias && q_out && workspace && temperature, "null pointer");
    SN_REQUIRE(Kfc >= 4 && Kfc % 4 == 0, "Kfc must be a multiple of 4");
    SN_REQUIRE(p_layout == SN_LAYOUT_BNC || p_layout == SN_LAYOUT_BCN, "bad p_layout");
    const int G = sn_pairscan_colmin_splits(B, N, M);
    if (G <= 1 || N > sn::kWave * 32)
        return sn_set_error(SN_ERR_UNSUPPORTED, "sn_pairscan_forward_partial_fc: this shape runs as one workgroup per cloud");
    SN_REQUIRE(workspace_bytes >= (long long)B * G * N * 8, "workspace too small");
    PairscanArgs a{};
    a.P = P, a.Q = q_out, a.p_layout = p_layout, a.q_layout = SN_LAYOUT_BCN;
    a.B = B, a.N = N, a.M = M, a.K = K;
    a.knn_idx = knn_idx, a.dist_q = dist_q, a.idx_q = idx_q;
    a.proj = proj, a.proj_layout = proj_layout, a.temperature = temperature, a.min_sigma = min_sigma;
    a.fc_z = fc_z, a.fc_scale = fc_scale, a.fc_shift = fc_shift, a.fc_w = fc_w, a.fc_bias = fc_bias, a.fc_k = Kfc;
    a.q_out = q_out;
    int used = 0;
    int rc = sn::pairscan_dispatch(a, workspace, workspace_bytes, false, &used, (hipStream_t)stream);
    if (rc) return rc;
    SN_LAUNCH_CHECK();  // (first: SN_REQUIRE discards the pending launch status)
    if (used != G) return sn_set_error(SN_ERR_BAD_ARGUMENT, "%s: internal: split mismatch", __func__);
    return 0;
}

#ifdef SN_PS_TIMELINE
// copies the first nblocks x 16 stamps of the last pair-scan launch to the host and clears them
extern "C" int sn_debug_pairscan_timeline(void *host, int nblocks)
{
    if (nblocks < 0 || nblocks > 8192) return -1;
    if (hipDeviceSynchronize() != hipSuccess) return -2;
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(sn::g_ps_tl), (size_t)nblocks * 16 * 8) != hipSuccess) return -3;
    static unsigned long long zeros[8192 * 16];
    return hipMemcpyToSymbol(HIP_SYMBOL(sn::g_ps_tl), zeros, sizeof(zeros)) == hipSuccess ? 0 : -4;
}
#endif
