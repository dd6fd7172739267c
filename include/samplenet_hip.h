/*
 * samplenet_hip.h -- C ABI of libsamplenet_hip.so: the MI355X (gfx950) implementation of the
 * SampleNet differentiable-sampling hot path.
 *
 * Conventions (all entry points):
 *   - every pointer is a DEVICE pointer to caller-owned memory; the library allocates nothing
 *     and keeps no state between calls (re-entrant; any number of streams / host threads);
 *   - tensors are dense fp32 / int32 in the layout stated per argument -- the same layouts the
 *     reference launchers take; no strides except the explicit `layout` selectors below;
 *   - `stream` is a hipStream_t (passed as void*); work is enqueued on it and the call returns
 *     without synchronising (safe under hipGraph stream capture);
 *   - return value: 0 on success, otherwise a hipError_t value or SN_ERR_*; the text of the
 *     most recent error of the calling thread is returned by sn_last_error_string().
 *
 * Each function names the reference interface it replaces (file:line under the
 * itailang/SampleNet checkout).  THIS header is the drop-in boundary: the entries a binding of the
 * reference's native interfaces (the `cd` pybind module, the TF op launchers, knn_cuda / pointnet2,
 * the torch.nn layer chain of samplenet.py:90-104) would call.  The fused-step entry points that the
 * host side of this package (samplenet_amd/{pointnet,fused_step,engine}.py) uses on top of them --
 * multi-layer launches, the sampler step's single-node loss, their scratch-size queries -- are
 * declared in samplenet_hip_internal.h; they carry no stability promise.
 */
#ifndef SAMPLENET_HIP_H
#define SAMPLENET_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define SN_ABI_VERSION 1

/* error codes beyond hipError_t (which are < 10000) */
#define SN_ERR_BAD_ARGUMENT 10001 /* null pointer, negative size, K > N, unsupported K ... */
#define SN_ERR_UNSUPPORTED 10002  /* shape outside what the kernels implement            */

/* point-cloud layout selectors */
#define SN_LAYOUT_BNC 0 /* (B, N, 3)  point-major  -- Chamfer / TF ops           */
#define SN_LAYOUT_BCN 1 /* (B, 3, N)  channel-major -- torch SoftProjection/KNN  */

typedef void *sn_stream_t;

int sn_abi_version(void);
const char *sn_last_error_string(void);
/* bytes of scratch sn_* functions with a `workspace` argument need for the given sizes */
long long sn_workspace_bytes(const char *op, int B, int N, int M, int K);

/* ---------------------------------------------------------------------------------------------
 * Fused pair scan: ONE pass over the M x N squared-distance matrix of every cloud producing
 *   - kNN indices of every query among the dataset points, ascending by (distance, index)
 *   - nearest dataset point of every query   (dist_q, idx_q)   [Chamfer direction 1]
 *   - nearest query of every dataset point   (dist_p, idx_p)   [Chamfer direction 2]
 *   - the soft projection of every query onto its K neighbours
 * Any output pointer may be NULL to skip that product.
 *
 *   P (dataset, N points) / Q (queries, M points): layout selectors above.
 *   knn_idx (B,M,K) int32, knn_dist (B,M,K) squared distances.
 *   dist_q/idx_q (B,M); dist_p/idx_p (B,N).
 *   proj: soft projection, written in `proj_layout`; weights (B,M,K) optional softmax weights.
 *   temperature: device pointer to the scalar T; sigma = max(T*T, min_sigma)
 *                (registration/src/soft_projection.py:97-99).
 * Replaces, in one launch: knn_cuda.KNN (call sites soft_projection.py:11-14, samplenet.py:121;
 * in-tree definition classification/grouping/tf_grouping.py:64-91 + tf_grouping_g.cu:83-123),
 * pointnet2 grouping_operation + the softmax/weighted-sum torch ops of soft_projection.py:138-152,
 * and ChamferDistanceKernelLauncher (chamfer_distance.cu:139-155).
 * Squared distance = ((dx*dx + dy*dy) + dz*dz) in fp32, one rounding per operation, never
 * contracted to FMA -- the expression of chamfer_distance.cpp:74-77.
 * ------------------------------------------------------------------------------------------- */
int sn_pairscan_forward(int B, int N, int M, int K,
                        const float *P, int p_layout, const float *Q, int q_layout,
                        int *knn_idx, float *knn_dist,
                        float *dist_q, int *idx_q, float *dist_p, int *idx_p,
                        float *proj, int proj_layout, float *weights,
                        const float *temperature, float min_sigma,
                        sn_stream_t stream);

/* Same, with caller-owned scratch (sn_pairscan_workspace_bytes(B,N,M) bytes): lets the queries of one cloud be
 * spread over several workgroups even when the per-point minima (dist_p / idx_p) are requested -- at the
 * reference's batch of 32 clouds that is what fills the 256 CUs.  workspace == NULL behaves like sn_pairscan_forward. */
long long sn_pairscan_workspace_bytes(int B, int N, int M);
int sn_pairscan_forward_ws(int B, int N, int M, int K,
                           const float *P, int p_layout, const float *Q, int q_layout,
                           int *knn_idx, float *knn_dist,
                           float *dist_q, int *idx_q, float *dist_p, int *idx_p,
                           float *proj, int proj_layout, float *weights,
                           const float *temperature, float min_sigma,
                           void *workspace, long long workspace_bytes, sn_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Chamfer / nn_distance with the reference launcher signatures.
 * Replaces ChamferDistanceKernelLauncher / ChamferDistanceGradKernelLauncher
 * (registration/src/chamfer_distance/chamfer_distance.cpp:4-24, .cu:139-209) and
 * NmDistanceKernelLauncher / NmDistanceGradKernelLauncher
 * (classification/structural_losses/tf_nndistance_g.cu:129-157).
 * xyz (b,n,3), xyz2 (b,m,3); result/result_i (b,n); result2/result2_i (b,m).
 * Backward: grad_xyz1 (b,n,3), grad_xyz2 (b,m,3) are fully overwritten (no memset needed);
 * either may be NULL to skip it.  Deterministic (no atomics).
 * ------------------------------------------------------------------------------------------- */
int sn_chamfer_forward(int b, int n, const float *xyz, int m, const float *xyz2,
                       float *result, int *result_i, float *result2, int *result2_i,
                       sn_stream_t stream);
int sn_chamfer_backward(int b, int n, const float *xyz1, int m, const float *xyz2,
                        const float *grad_dist1, const int *idx1,
                        const float *grad_dist2, const int *idx2,
                        float *grad_xyz1, float *grad_xyz2, sn_stream_t stream);

/* Progressive sampler (classification/train_samplenet_progressive.py:194-234: the nested prefixes Q[:, :s_j] of ONE simplified
 * cloud each take the simplification loss): the per-POINT side of every prefix's Chamfer products from one pass over the
 * M x N distances.  P (B,N,3), Q (B,M,3); prefix_sizes[nprefix] HOST array, ascending, last == M, nprefix <= 16;
 * dist / idx: (nprefix, B, N) -- slice j equals sn_chamfer_forward(Q[:, :s_j], P)'s dist2 / idx2 bit for bit.
 * (The per-query side does not depend on the prefix: slice dist1 / idx1 of the full scan.) */
int sn_prefix_point_minima(int B, int N, int M, int nprefix, const int *prefix_sizes, const float *P, const float *Q,
                           float *dist, int *idx, sn_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Fused simplification loss of the sampler (registration/src/samplenet.py:171-181; TF twin
 * classification/models/samplenet_model.py:176-188):
 *     loss = mean(dist1) + mean_b(max_m dist1[b,m]) + weight * mean(dist2),   weight = gamma + delta * pc_size
 * dist1 (B,n1) / dist2 (B,n2) are the Chamfer products of (xyz1 = sampled cloud, xyz2 = reference cloud).
 * forward: partial (B*3 floats) and argmax1 (B ints) are caller-owned scratch kept for backward; loss: 1 float.
 * backward: grad_loss is a DEVICE scalar; writes grad_xyz1 (B,n1,3) / grad_xyz2 (B,n2,3) (either may be NULL) without
 * materialising per-point gradient tensors.  Replaces ~10 elementwise/reduction launches + the Chamfer backward.
 * ------------------------------------------------------------------------------------------- */
int sn_simplification_loss_forward(int B, int n1, int n2, const float *dist1, const float *dist2, float weight,
                                   float *partial, int *argmax1, float *loss, sn_stream_t stream);
int sn_simplification_loss_backward(int B, int n1, const float *xyz1, int n2, const float *xyz2, const int *idx1,
                                    const int *idx2, const int *argmax1, float weight, const float *grad_loss,
                                    float *grad_xyz1, float *grad_xyz2, int layout1, sn_stream_t stream);
/* layout1: 0 = xyz1 / grad_xyz1 are (B,n1,3); 1 = (B,3,n1), the layout the sampler's FC head emits (grad_xyz2 must be NULL) */

/* ---------------------------------------------------------------------------------------------
 * The registration task's own loss pieces (registration/main.py:557-577, compute_pcrnet_loss), one launch pair each:
 *   sn_chamfer_mean_loss_*   chamfer_loss = mean(dist1) + mean(dist2) from the Chamfer products of (xyz1, xyz2) (main.py:573-577)
 *                            and its gradients to either cloud (grad_loss: device scalar; either output may be NULL) -- the
 *                            simplification loss without the maximum term.  partial: 3 B floats, argmax1: B ints of scratch.
 *   sn_pcrnet_head_*         twist (B,7) = [normalize(y[:, 0:4]) | y[:, 4:7]] (models/pcrnet.py:78-82, F.normalize eps 1e-12) and
 *                            qnorm = mean_b (||y[:, 0:4]||^2 - 1)^2 (main.py:565; NULL: not wanted); quat (B,4), optional: the
 *                            normalised quaternion once more as a contiguous tensor (what QuaternionTransform.rotate consumes:
 *                            no slice / copy launches around it); backward: g_y (B,7) from g_twist (B,7), g_quat (B,4) and the
 *                            device scalar g_qnorm (each may be NULL).
 * ------------------------------------------------------------------------------------------- */
int sn_chamfer_mean_loss_forward(int B, int n1, int n2, const float *dist1, const float *dist2, float *partial, int *argmax1,
                                 float *loss, sn_stream_t stream);
int sn_chamfer_mean_loss_backward(int B, int n1, const float *xyz1, int n2, const float *xyz2, const int *idx1, const int *idx2,
                                  const float *grad_loss, float *grad_xyz1, float *grad_xyz2, sn_stream_t stream);
int sn_pcrnet_head_forward(int B, const float *y, float *twist, float *quat, float *qnorm, sn_stream_t stream);
int sn_pcrnet_head_backward(int B, const float *y, const float *g_twist, const float *g_quat, const float *g_qnorm, float *g_y,
                            sn_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * kNN alone (no gradient).  xyz1 dataset, xyz2 queries, layouts selectable.
 * Replaces knn_cuda.KNN(k)(ref, query) (soft_projection.py:11-14) and knn_point
 * (tf_grouping.py:64-91).  idx (b,m,k) int32; dist (b,m,k) SQUARED distances (may be NULL).
 * ------------------------------------------------------------------------------------------- */
int sn_knn(int b, int n, int m, int k, const float *xyz1, int layout1, const float *xyz2, int layout2,
           int *idx, float *dist, sn_stream_t stream);

/* Inference matching on the device (registration/src/sputils.py:7-41 nn_matching / _fps_from_given_pc / _unique, called
 * from samplenet.py:119-141): idx (B,k) = nearest input point of every generated point; complete_fps != 0: first
 * occurrences in order, then farthest-point completion to k points with numpy's float64 distances and first-maximum
 * argmax -- exact index parity.  xyz: (B,N,3) or (B,3,N) by layout; out (B,k,3).  k <= 1024, N <= 8192. */
int sn_nn_matching(int B, int N, int k, const float *xyz, int layout, const int *idx, int complete_fps, float *out,
                   sn_stream_t stream);

/* Rotation of every cloud by its own quaternion: out[b][n] = qrot(quat[b], v[b][n]) with quat (B,4) in
 * (w, x, y, z) order, v / out (B,N,3) -- registration/src/quaternion.py:35-53 (qrot) as registration/main.py:569-571 uses it
 * through QuaternionTransform.rotate (qdataset.py:106-109: the quaternion expanded over the points).  backward: grad_v (B,N,3)
 * and grad_quat (B,4) (either may be NULL); deterministic (fixed-order per-cloud sums).  One launch each instead of ~8 / ~16. */
int sn_qrot_forward(int B, int N, const float *quat, const float *v, float *out, sn_stream_t stream);
int sn_qrot_backward(int B, int N, const float *quat, const float *v, const float *grad_out, float *grad_quat, float *grad_v,
                     sn_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * group_point gather / scatter-add.
 * (a) TF layout: points (b,n,c), idx (b,m,nsample) -> out (b,m,nsample,c).
 *     Replaces groupPointLauncher / groupPointGradLauncher (tf_grouping_g.cu:133-141;
 *     op shells tf_grouping.cpp:142-208).  grad_points is overwritten (zero-filled inside).
 * (b) channel-major: features (b,c,n), idx (b,m,nsample) -> out (b,c,m,nsample).
 *     Replaces pointnet2_utils.grouping_operation fwd/bwd (call site soft_projection.py:83-89).
 * ------------------------------------------------------------------------------------------- */
int sn_group_point(int b, int n, int c, int m, int nsample, const float *points, const int *idx,
                   float *out, sn_stream_t stream);
int sn_group_point_grad(int b, int n, int c, int m, int nsample, const float *grad_out,
                        const int *idx, float *grad_points, sn_stream_t stream);
int sn_grouping_operation(int b, int c, int n, int m, int nsample, const float *features,
                          const int *idx, float *out, sn_stream_t stream);
int sn_grouping_operation_grad(int b, int c, int n, int m, int nsample, const float *grad_out,
                               const int *idx, float *grad_features, sn_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * SoftProjection pieces (channel-major tensors, as the torch module uses them).
 *   sn_soft_weights_forward : w = softmax_k(-|P[idx]-Q|^2 / sigma)          soft_projection.py:92-95,143
 *   sn_soft_weights_backward: grad_w -> grad_Q (b,3,m) [overwritten], grad_P (b,3,n)
 *                             [ACCUMULATED with atomics; may be NULL], grad_sigma_partial (b)
 *                             [overwritten; sum it and apply d sigma/dT on the caller side]
 *   sn_weighted_gather_forward : out[c,j] = sum_k w[j,k] X[c, idx[j,k]]      soft_projection.py:113-118,131-134,148-151
 *   sn_weighted_gather_backward: grad_out -> grad_w (b,m,k) [overwritten, may be NULL],
 *                                grad_X (b,c,n) [ACCUMULATED with atomics, may be NULL]
 *   sn_soft_project_backward   : fused backward of `project` for the hot path:
 *                                grad_proj (layout selectable) -> grad_Q (b,3,m) overwritten,
 *                                grad_sigma_partial (b * sn_soft_bwd_splits(b,m)) overwritten, grad_P optional
 *                                (atomics).  sn_soft_weights_backward: same partial count.
 * ------------------------------------------------------------------------------------------- */
int sn_soft_bwd_splits(int b, int m);
/* d loss/dT = (sum of the grad_sigma partials) * d max(T^2, min_sigma)/dT  (soft_projection.py:97-99), one launch */
int sn_sigma_grad(int nparts, const float *partial, const float *temperature, float min_sigma, float *grad_T,
                  sn_stream_t stream);
int sn_soft_weights_forward(int b, int n, int m, int k, const float *P, const float *Q, const int *idx,
                            const float *temperature, float min_sigma, float *weights,
                            sn_stream_t stream);
int sn_soft_weights_backward(int b, int n, int m, int k, const float *P, const float *Q, const int *idx,
                             const float *temperature, float min_sigma, const float *weights,
                             const float *grad_weights, float *grad_Q, float *grad_P,
                             float *grad_sigma_partial, sn_stream_t stream);
int sn_weighted_gather_forward(int b, int c, int n, int m, int k, const float *X, const int *idx,
                               const float *weights, float *out, sn_stream_t stream);
int sn_weighted_gather_backward(int b, int c, int n, int m, int k, const float *X, const int *idx,
                                const float *weights, const float *grad_out, float *grad_weights,
                                float *grad_X, sn_stream_t stream);
int sn_soft_project_backward(int b, int n, int m, int k, const float *P, int p_layout,
                             const float *Q, int q_layout, const int *idx,
                             const float *temperature, float min_sigma,
                             const float *grad_proj, int gproj_layout,
                             float *grad_Q, int gq_layout, float *grad_P, float *grad_sigma_partial,
                             sn_stream_t stream);
/* The same three backward entries with a DETERMINISTIC gradient towards the point cloud / the features: the contributions are
 * summed per destination point in the order of the reference's loops (query ascending, neighbour ascending;
 * soft_projection.py:75-152 through autograd's index_add on CPU) instead of by unordered float atomics -- bit-identical from
 * run to run and to the CPU oracle.  grad_P / grad_X (required) are OVERWRITTEN, not accumulated.  scratch: b * m * k * 3 floats
 * (sn_weighted_gather_backward_ordered: b * c * m * k). */
int sn_soft_project_backward_ordered(int b, int n, int m, int k, const float *P, int p_layout, const float *Q, int q_layout,
                                     const int *idx, const float *temperature, float min_sigma, const float *grad_proj,
                                     int gproj_layout, float *grad_Q, int gq_layout, float *grad_P,
                                     float *grad_sigma_partial, float *scratch, sn_stream_t stream);
int sn_soft_weights_backward_ordered(int b, int n, int m, int k, const float *P, const float *Q, const int *idx,
                                     const float *temperature, float min_sigma, const float *weights,
                                     const float *grad_weights, float *grad_Q, float *grad_P, float *grad_sigma_partial,
                                     float *scratch, sn_stream_t stream);
int sn_weighted_gather_backward_ordered(int b, int c, int n, int m, int k, const float *X, const int *idx,
                                        const float *weights, const float *grad_out, float *grad_weights, float *grad_X,
                                        float *scratch, sn_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * EMD: approx_match / match_cost / match_cost_grad.
 * Replaces approxmatchLauncher / matchcostLauncher / matchcostgradLauncher
 * (classification/structural_losses/tf_approxmatch_g.cu:180-182,226-228,292-295; op shells
 * tf_approxmatch.cpp:145-329).  xyz1 (b,n,3), xyz2 (b,m,3), match (b,m,n) (may be NULL: only the
 * per-level ratio vectors are then produced), cost (b), grad1 (b,n,3), grad2 (b,m,3).
 * temp: caller-owned scratch of sn_workspace_bytes("approxmatch", b, n, m, 0) bytes -- a superset
 * of the (b,(n+m)*2) floats the reference op allocates (tf_approxmatch.cpp:167-168): the ratio
 * vectors of all 10 levels are kept so that `match` is written once instead of being
 * read-modify-written per level.  sn_matchcost needs sn_workspace_bytes("matchcost", ...) bytes
 * of scratch for its per-workgroup partial sums (deterministic two-stage reduction).
 * ------------------------------------------------------------------------------------------- */
int sn_approxmatch(int b, int n, int m, const float *xyz1, const float *xyz2, float *match,
                   float *temp, sn_stream_t stream);
int sn_matchcost(int b, int n, int m, const float *xyz1, const float *xyz2, const float *match,
                 float *cost, float *workspace, sn_stream_t stream);
int sn_matchcost_grad(int b, int n, int m, const float *xyz1, const float *xyz2, const float *match,
                      float *grad1, float *grad2, sn_stream_t stream);
/* EMD loss in one call, without the (b,m,n) match matrix: cost (b) = match_cost(xyz1, xyz2, approx_match(xyz1, xyz2)) as
 * reconstruction/src/samplenet_pointnet_ae.py:129-131 composes it, and -- when grad1 (b,n,3) / grad2 (b,m,3) are given -- the
 * gradient of each cost w.r.t. its clouds with match held constant (tf_approxmatch.py:54-64).  match[l,k] is re-evaluated
 * from the per-level ratio vectors inside the sweeps: 839 MB never written at B = 50, 2048 x 2048.  cost and grad1 are
 * bit-identical to sn_matchcost / sn_matchcost_grad on the materialised match; grad2 sums in a different order (1e-5).
 * temp: sn_workspace_bytes("emd_loss", b, n, m, 0) bytes. */
int sn_emd_loss(int b, int n, int m, const float *xyz1, const float *xyz2, float *cost, float *grad1, float *grad2,
                float *temp, sn_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * PointNet feature extractor + FC head (registration/src/samplenet.py:40-59, :90-104): every layer is a
 * GEMM over rows (R = B*N points for the 1x1 convolutions, R = B for the Linear layers) on fp32 MFMA
 * with BatchNorm / ReLU / bias / reductions fused (samplenet_amd/csrc/pointnet_mlp.hip forward, pointnet_mlp_backward.hip, fc_chain.hip; the task network: task_network.hip).
 * All matrices are row-major with channels contiguous: activations [R][C], weights [Co][Ci] (the
 * memory layout of torch Conv1d(k=1).weight and Linear.weight).
 *
 *  sn_linear_forward   Z[R][Co] = act(Ain)[R][Ci] . W^T + bias;  act = relu(scale*x + shift) with
 *                      coef_prev = {scale[Ci], shift[Ci], ...} or identity when coef_prev == NULL.
 *                      stats (optional): [sn_linear_stats_blocks(R)][2][Co] partial sum / sum of squares.
 *  sn_bn_finalize      batch statistics -> coef[4][C] = scale, shift, mean, invstd; updates the running
 *                      statistics and num_batches_tracked like torch.nn.BatchNorm1d in training mode.
 *  sn_bn_eval_coef     coef from the running statistics (eval mode).
 *  sn_pool_forward     pooled[b][c] = max_n relu(bn(z[b][n][c])) (+ index and pre-BN value of the selected point).
 *  sn_pool_backward    gsel = g * [pooled > 0] and the BN-backward sums of the last conv layer.
 *  sn_bn_backward_coef (sum dY, sum dY*Z) partials -> dgamma, dbeta, dbias, kcoef[3][C] with dZ = k1 dY + k2 Z + k3.
 *  sn_linear_dgrad     dYprev[R][Ci] = relu-mask . (dZ[R][Co] . W), + BN-backward partial sums of the previous
 *                      layer [sn_linear_stats_blocks(R)][2][Ci];  dz_mode 0: dZ = dy; 1: k1 dy + k2 z + k3;
 *                      2: as 1 with dy given sparsely by (gsel, argsel) per cloud of npts rows.
 *  sn_linear_wgrad     dW[Co][Ci] (and db[Co] if non-NULL) = dZ^T . act(Aprev); `part` is scratch of
 *                      sn_linear_wgrad_splits(R,Ci,Co,db!=NULL) * Co * (Ci + (db!=NULL)) floats.
 * ------------------------------------------------------------------------------------------- */
int sn_linear_stats_blocks(int R);
int sn_linear_forward(int R, int Ci, int Co, const float *ain, const float *coef_prev, const float *W,
                      const float *bias, float *z, float *stats, sn_stream_t stream);
int sn_bn_finalize(int nblk, int C, long long R, const float *stats, const float *gamma, const float *beta,
                   float eps, float momentum, float *running_mean, float *running_var,
                   long long *num_batches_tracked, float *coef, sn_stream_t stream);
int sn_bn_eval_coef(int C, const float *gamma, const float *beta, float eps, const float *running_mean,
                    const float *running_var, float *coef, sn_stream_t stream);
int sn_pool_forward(int B, int N, int C, const float *z, const float *coef, float *pooled, int *argsel,
                    float *zsel, sn_stream_t stream);
int sn_pool_backward(int B, int C, const float *g, const float *pooled, const float *zsel, float *gsel,
                     float *stats, sn_stream_t stream);
int sn_bn_backward_coef(int nblk, int C, long long R, const float *stats, const float *coef, float *dgamma,
                        float *dbeta, float *dbias, float *kcoef, sn_stream_t stream);
int sn_linear_dgrad(int R, int Ci, int Co, int dz_mode, const float *dy, const float *z, const float *kcoef,
                    const float *gsel, const int *argsel, int npts, const float *W, const float *zprev,
                    const float *coef_prev, float *dyprev, float *stats, sn_stream_t stream);
int sn_linear_wgrad_splits(int R, int Ci, int Co, int with_bias);
int sn_linear_wgrad(int R, int Ci, int Co, int dz_mode, const float *dy, const float *z, const float *kcoef,
                    const float *gsel, const int *argsel, int npts, const float *aprev, const float *coef_prev,
                    float *part, float *dW, float *db, sn_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SAMPLENET_HIP_H */
