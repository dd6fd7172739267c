/*
 * samplenet_hip_internal.h -- entry points of libsamplenet_hip.so BEHIND the drop-in boundary: the fused
 * multi-layer / single-node forms of the public operations (include/samplenet_hip.h) that the package's own host
 * side drives (samplenet_amd/pointnet.py, fused_step.py, engine.py).  Same calling conventions as the public
 * header (caller-owned device buffers, stream-ordered, 0 / error code).  None of these replaces a reference
 * interface one to one -- each names the public entries / reference lines whose composition it computes -- and
 * they may change with the kernels (no ABI promise: sn_abi_version() covers samplenet_hip.h only).
 */
#ifndef SAMPLENET_HIP_INTERNAL_H
#define SAMPLENET_HIP_INTERNAL_H

#include "samplenet_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* The sampler's total loss with the benchmark's stand-in task term (registration/main.py:507-531):
 *     L = alpha * L_simp + lmbda * max(T^2, min_sigma) + mean(proj)        (all operands device scalars / tensors)
 * forward: loss (1 float).  backward: grad_proj (nproj floats, = grad_loss / nproj), grad_lsimp (1), grad_T (1). */
int sn_sampler_loss_forward(int nproj, const float *proj, const float *lsimp, const float *temperature,
                            float alpha, float lmbda, float min_sigma, float *loss, sn_stream_t stream);
int sn_sampler_loss_backward(int nproj, const float *grad_loss, const float *temperature, float alpha, float lmbda,
                             float min_sigma, float *grad_proj, float *grad_lsimp, float *grad_T, sn_stream_t stream);

/* The sampler's training-step loss with the benchmark's stand-in task term, in the fewest launches the dependencies allow
 * (the op-by-op route above computes the same numbers):
 *   sn_pairscan_forward_partial   pair scan that leaves the per-point minima as G = sn_pairscan_colmin_splits(B,N,M) partial
 *                                 key sets (workspace [B][G][N] u64); SN_ERR_UNSUPPORTED when G <= 1
 *   sn_sampler_step_loss_forward  finishes dist_p / idx_p from those partials while reducing the loss; loss[0] = L,
 *                                 loss[1] = L_simp; partial: B*4 floats; argmax1: B ints
 *   sn_sampler_step_loss_backward grad_Q (B,3,M) = d L / d simplified cloud, grad_T; gsig_scratch: B*sn_soft_bwd_splits(B,M) */
int sn_pairscan_colmin_splits(int B, int N, int M);
int sn_pairscan_forward_partial(int B, int N, int M, int K, const float *P, int p_layout, const float *Q, int q_layout,
                                int *knn_idx, float *dist_q, int *idx_q, float *proj, int proj_layout,
                                const float *temperature, float min_sigma, void *workspace, long long workspace_bytes,
                                sn_stream_t stream);
/* sn_pairscan_forward_partial with the queries PRODUCED inside the scan by the head's last fully connected layer
 * (samplenet.py:103-104): q[b][c][j] = fc_bias[c*M+j] + sum_k relu(fc_z[b][k]*fc_scale[k] + fc_shift[k]) * fc_w[c*M+j][k];
 * q_out (B,3,M) receives the simplified cloud.  Kfc % 4 == 0.  One launch less than FC layer + scan. */
int sn_pairscan_forward_partial_fc(int B, int N, int M, int K, const float *P, int p_layout, const float *fc_z,
                                   const float *fc_scale, const float *fc_shift, const float *fc_w, const float *fc_bias,
                                   int Kfc, float *q_out, int *knn_idx, float *dist_q, int *idx_q, float *proj,
                                   int proj_layout, const float *temperature, float min_sigma, void *workspace,
                                   long long workspace_bytes, sn_stream_t stream);
int sn_sampler_step_loss_forward(int B, int M, int N, int G, const float *dist_q, const void *colmin_ws, const float *proj,
                                 const float *temperature, float alpha, float lmbda, float weight, float min_sigma,
                                 float *dist_p, int *idx_p, int *argmax1, float *partial, float *loss, int defer_value,
                                 sn_stream_t stream);
/* sn_sampler_step_loss_forward behind a scan that ran one workgroup per cloud (sn_pairscan_colmin_splits <= 1: batches that
 * fill the chip on their own; dist_p / idx_p complete, e.g. from sn_pairscan_forward_ws): same partial / loss outputs. */
int sn_sampler_step_loss_forward_direct(int B, int M, int N, const float *dist_q, const float *dist_p, const float *proj,
                                        const float *temperature, float alpha, float lmbda, float weight, float min_sigma,
                                        int *argmax1, float *partial, float *loss, int defer_value, sn_stream_t stream);
int sn_sampler_step_loss_backward(int B, int N, int M, int K, const float *P, int p_layout, const float *Q,
                                  const int *knn_idx, const int *idx_q, const int *idx_p, const int *argmax1,
                                  const float *temperature, float min_sigma, float alpha, float lmbda, float weight,
                                  const float *grad_loss, float *grad_Q, float *gsig_scratch, float *grad_T,
                                  const float *deferred_partial, float *deferred_loss, sn_stream_t stream);
/* The same step with NO launch between the scan and the backward (engine default): sn_pairscan_forward_keys combines the
 * per-point minima across a cloud's workgroups by 64-bit atomicMax on inverted (distance, query) keys (max / min are
 * order-independent: deterministic) in colmin_keys [B][N] -- u64, zero on entry, zero again after
 * sn_sampler_step_loss_keys -- and leaves the query-side loss partials qpart [B][G][2] floats / qmax [B][G] u64,
 * G = sn_pairscan_colmin_splits(B,N,M) > 1.  Q (B,3,M) is read, or written when fc_w != NULL (queries produced by the
 * head's last layer as in sn_pairscan_forward_partial_fc).  dpsum: B floats of scratch; loss: 2 floats.  N <= 2048.
 * deferred_tail (optional, host buffer of sn_step_tail_bytes() bytes): the launch that produces grad_T / loss and resets
 * colmin_keys is not issued at all; its description is written there and handed to sn_conv_stack_backward(step_tail) of the same step, whose closing kernel runs it. */
int sn_pairscan_forward_keys(int B, int N, int M, int K, const float *P, int p_layout, float *Q, const float *fc_z,
                             const float *fc_scale, const float *fc_shift, const float *fc_w, const float *fc_bias, int Kfc,
                             int *knn_idx, float *dist_q, int *idx_q, float *proj, int proj_layout, const float *temperature,
                             float min_sigma, void *colmin_keys, float *qpart, void *qmax, sn_stream_t stream);
int sn_sampler_step_loss_keys(int B, int N, int M, int K, const float *P, int p_layout, const float *Q, const int *knn_idx,
                              const int *idx_q, void *colmin_keys, const float *qpart, const void *qmax, int G,
                              const float *temperature, float min_sigma, float alpha, float lmbda, float weight,
                              const float *grad_loss, float *grad_Q, float *gsig_scratch, float *grad_T, float *dpsum,
                              float *loss, sn_stream_t stream, void *deferred_tail, const float *grad_proj,
                              const float *grad_sigma);
/* grad_proj (B,M,3), optional: gradient of a task loss that lives OUTSIDE this node w.r.t. the projected points (the task
 * network of registration/main.py:507-531 sits on proj); the loss value then carries no mean(proj) term.  NULL: the
 * benchmark's stand-in term mean(proj) is part of the loss and its gradient grad_loss / (3 B M) is implicit. */
/* grad_sigma (1 float, optional): upstream gradient of sigma = max(T^2, min_sigma) as an OUTPUT of the caller's node (the
 * drop-in surface: the script forms lmbda * get_projection_loss() itself); grad_T's direct term is then grad_sigma * d sigma / dT
 * instead of lmbda * grad_loss * d sigma / dT. */
/* The drop-in module surface on captured work (samplenet_amd/surface.py; registration/main.py:507-531 calls net(x), the two
 * loss getters and backward() itself): values of L_simp and sigma right behind sn_pairscan_forward_keys -- values[0] = L_simp
 * (samplenet.py:171-181 with `weight` = gamma + delta * pc_size), [1] = sigma, [2..4] = mean dist_q, mean_b max_m dist_q,
 * mean dist_p; simp_bnc (optional): y_bcn (B,3,M) transposed to (B,M,3); dpsum: B floats of scratch; the key table is read,
 * not reset.  sn_surface_gather_upstream: the node's three upstream gradients (each optional: NULL = zero) into the static
 * operands of the captured backward: scalars[0] = d / d L_simp, scalars[1] = d / d sigma, proj_out (nproj floats). */
int sn_surface_values_keys(int B, int N, int M, int G, const void *colmin_keys, const float *qpart, const void *qmax,
                           const float *temperature, float min_sigma, float weight, const float *y_bcn, float *simp_bnc,
                           float *dpsum, float *values, sn_stream_t stream);
int sn_surface_gather_upstream(int nproj, const float *g_lsimp, const float *g_sigma, const float *g_proj, float *scalars,
                               float *proj_out, sn_stream_t stream);
/* From how many 64-row tiles per workgroup sn_conv_stack_forward_bn runs its GEMM layers as persistent weight-stationary
 * kernels (large batches; default 4, 0: never).  Returns the previous value.  A test / A-B hook: results are bit-identical. */
int sn_conv_stack_set_persist_min_tiles(int tiles);
int sn_step_tail_bytes(void);
/* Names the error words of the step's FC chain launches (the `sync` buffers of sn_fc_chain_forward[_pool] /
 * sn_fc_chain_backward, either may be NULL) in a deferred-tail blob: the loss value the tail writes is NaN when a hand-off
 * of one of those launches timed out (sync[15] != 0). */
int sn_step_tail_set_error_words(void *tail_blob, const void *fwd_sync, const void *bwd_sync);
/* defer_value != 0: the forward leaves loss[] unwritten; pass its `partial` and `loss` to the backward call as
 * deferred_partial / deferred_loss and the scalar is combined by an extra wave of the backward's first launch (the
 * gradients do not depend on it) -- for callers that always run the backward (samplenet_amd.engine).  Else pass NULLs. */


/* layer-level entry points (what samplenet_amd/pointnet.py calls): a layer's forward INCLUDING its BatchNorm
 * finalisation, and a layer's backward (dW, optional db, dYprev) INCLUDING the BatchNorm backward coefficients of the
 * layer below; they pick the fused single-/dual-launch kernels by shape and fall back to the pieces above otherwise. */
int sn_layer_forward_bn(int R, int Ci, int Co, const float *ain, const float *coef_prev, const float *W,
                        const float *bias, float *z, float *stats, const float *gamma, const float *beta,
                        float eps, float momentum, float *running_mean, float *running_var,
                        long long *num_batches_tracked, float *coef, sn_stream_t stream);
/* sn_layer_forward_bn of the last conv layer with the max-pool over the npts points of every cloud folded in (the forward
 * epilogue leaves per-block maxima / minima, the BatchNorm finalisation picks): pooled / argsel / zsel as sn_pool_forward;
 * pool_val / pool_idx: scratch of sn_linear_stats_blocks(R) * 2 * Co elements each.  SN_ERR_UNSUPPORTED unless R, npts, Ci,
 * Co are multiples of 64. */
int sn_conv_forward_bn_pool(int R, int Ci, int Co, int npts, const float *ain, const float *coef_prev, const float *W,
                            const float *bias, float *z, float *stats, const float *gamma, const float *beta, float eps,
                            float momentum, float *running_mean, float *running_var, long long *num_batches_tracked,
                            float *coef, float *pool_val, int *pool_idx, float *pooled, int *argsel, float *zsel,
                            sn_stream_t stream);
/* The training-mode conv stack (conv/bn/relu x nlayers on the xyz cloud + max over the points, samplenet.py:90-95) in one
 * call and nlayers + 1 launches.  Batch statistics travel as 64-bit fixed-point sums accumulated with integer atomics
 * (order-independent, hence deterministic); each layer finalises the BatchNorm of its input itself, so no reduction launch
 * sits between layers.  channels [nlayers+1] = 3, C1..Cn (64 or 128 each; inner layers up to 256 --
 * reconstruction/src/samplers.py:23-38 -- when pooled != NULL and B * N / 64 row blocks can carry the weight split); N % 64 == 0
 * (query _supported first).
 * Arrays of nlayers device pointers: W (C_{l+1},C_l), bias, gamma, beta, running_mean, running_var, num_batches_tracked,
 * z (B*N,C_{l+1}) pre-BN outputs, coef (4,C_{l+1}); eps / momentum: host arrays.  acc: sn_conv_stack_acc_elems(nlayers) long long of persistent
 * device scratch, zero before the first call (every call leaves it zero).  pool_val / pool_idx: (B*N/64)*2*Cn scratch.
 * pooled = argsel = zsel = NULL: stop after the last GEMM; sn_fc_chain_forward_pool must follow (it finishes the pool).  The last
 * layer then leaves, instead of block partials, (B, 2, Cn) 64-bit keys in pool_val -- per cloud and channel (max Z, first row)
 * and (min Z, first row), combined by atomicMax -- which sn_fc_chain_forward_pool decodes; pool_val must then hold at least
 * 4 * B * Cn floats (16 bytes per cloud and channel: more than the block partials only when N < 128). */
int sn_conv_stack_forward_supported(int B, int N, int nlayers, const int *channels);
long long sn_conv_stack_acc_elems(int nlayers);
/* 1: z[0] may be NULL in sn_conv_stack_forward_bn / sn_conv_stack_backward for this shape -- the xyz layer then runs as a
 * statistics-only pass and its activation (B*N, C1) is never written: conv2's forward and backward rebuild it from the cloud
 * with the xyz layer's own expression (bit-identical results, 8 MB less written and 16 MB less read per step at B = 32). */
int sn_conv_stack_z1_free_supported(int B, int N, int nlayers, const int *channels);
/* the leading part of acc that holds the statistics accumulators (zero between calls): one block of sums per layer, two when a
 * layer of the stack is wider than 128 channels (channels: the nlayers + 1 widths as passed to sn_conv_stack_forward_bn, or NULL for
 * the narrow layout); behind it: scratch of the forward (the layers' weights split into bf16 planes by the first kernel of the
 * call), which is not zero between calls */
long long sn_conv_stack_acc_sum_elems(int nlayers, const int *channels);
int sn_conv_stack_forward_bn(int B, int N, int nlayers, const int *channels, const float *x, const float *const *W,
                             const float *const *bias, const float *const *gamma, const float *const *beta,
                             float *const *running_mean, float *const *running_var, long long *const *num_batches_tracked,
                             const float *eps, const float *momentum, float *const *z, float *const *coef, long long *acc,
                             float *pool_val, int *pool_idx, float *pooled, int *argsel, float *zsel, sn_stream_t stream);
/* Backward of the conv stack, the mirror of sn_conv_stack_forward_bn, in nlayers launches: one fused dgrad + wgrad kernel
 * per GEMM layer (top first) and one closing kernel that reduces every layer's weight-gradient partials and finishes the
 * xyz layer (BatchNorm backward, closed-form weight gradient).  The BatchNorm-backward sums travel between the kernels as
 * fixed-point atomics; each kernel derives its own layer's dZ coefficients in its prologue.
 * gsel / argsel (B,Cn), kcoef_top (3,Cn): pooled gradient at the selected points and the top BatchNorm's dZ coefficients
 * (sn_layer_backward with prev_bn_rows, or sn_pool_backward_bn).  dW: nlayers outputs; dgamma / dbeta / dbias: outputs for
 * layers 0 .. nlayers-2.  acc: sn_conv_stack_acc_elems(nlayers) long long of persistent zeroed scratch (its own buffer, not
 * the forward's); scratch: sn_conv_stack_backward_scratch_floats(...) floats (0 = shape not supported). */
long long sn_conv_stack_backward_scratch_floats(int B, int N, int nlayers, const int *channels);
int sn_conv_stack_backward(int B, int N, int nlayers, const int *channels, const float *x, const float *const *W,
                           const float *bias0, const float *const *z, const float *const *coef, const float *gsel,
                           const int *argsel, const float *kcoef_top, long long *acc, float *scratch, float *const *dW,
                           float *const *dgamma, float *const *dbeta, float *const *dbias, const void *step_tail,
                           sn_stream_t stream);
int sn_layer_backward(int R, int Ci, int Co, int dz_mode, const float *dy, const float *z, const float *kcoef,
                      const float *gsel, const int *argsel, int npts, const float *W, const float *zprev,
                      const float *coef_prev, float *dyprev, float *stats, float *part, float *dW, float *db,
                      float *prev_dgamma, float *prev_dbeta, float *prev_dbias, float *prev_kcoef, long long prev_bn_rows,
                      sn_stream_t stream);
/* sn_layer_backward (dz_mode = SN_DZ_BN) for the layer that sits on the xyz input layer (3 -> Ci, x_in (R,3), W_in (Ci,3),
 * b_in (Ci) or NULL): also returns dW_in (Ci,3), the input layer's weight gradient, in closed form from three extra
 * per-channel sums and the second moments of x_in accumulated by the same kernel -- no separate pass over dYprev, which
 * is not written at all (the input layer has no gradient to pass further down).
 * stats: sn_layer_backward_in3_stats_floats(R, Ci, Co) floats; that function returns 0 when the shape is not supported. */
long long sn_layer_backward_in3_stats_floats(int R, int Ci, int Co);
int sn_layer_backward_in3(int R, int Ci, int Co, const float *dy, const float *z, const float *kcoef, const float *W,
                          const float *zprev, const float *coef_prev, float *stats, float *part, float *dW,
                          float *prev_dgamma, float *prev_dbeta, float *prev_dbias, float *prev_kcoef, const float *x_in,
                          const float *W_in, const float *b_in, float *dW_in, sn_stream_t stream);
/* prev_bn_rows (> 0: R <= 32 only; 0 = R): rows seen by the BatchNorm of the layer below when they differ from R -- the FC head's
 * first layer on top of the max-pool: zprev = pooled pre-BN values (B rows), BatchNorm over B*N rows; replaces sn_pool_backward.
 * prev_bn_rows < 0 (any R): that BatchNorm normalised with FIXED statistics (eval mode: coef_prev from sn_bn_eval_coef), so its
 * backward is dZ = scale * dY (k2 = k3 = 0); dgamma / dbeta as usual.  Likewise R < 0 in sn_pool_backward_bn. */

/* The FC head's BatchNorm + ReLU layers (R <= 32 rows; a0 (R, C0) -> H -> ... -> H, nl layers) as ONE launch: the H / 32
 * workgroups of a layer stay resident and exchange each layer's activations through xbuf with write-through stores and an
 * arrival counter instead of ending the kernel per layer.  Per layer l: W[l] (H, K), bias, BatchNorm parameters / running
 * statistics (training-mode update as sn_layer_forward_bn), outputs z[l] (R, H) pre-BN and coef[l] (4, H).  Bit-identical to
 * nl calls of sn_layer_forward_bn.  xbuf: 2 * 32 * H floats of scratch; sync: 16 WORDS of state spaced 128 bytes apart (16 x 32 unsigned = 2 KB; "sync[i]"
 * below is the 32-bit word at byte offset 128 i: every counter on its own cache line), PERSISTENT and zero-initialised
 * once by the caller (epoch + monotonic arrival counters).  The workgroups of a launch must be RESIDENT TOGETHER (8 x ~137 KB
 * of LDS on otherwise free CUs).  A hand-off poll that gives up -- after sync[13] polls when that word is non-zero, else 2^22 --
 * sets sync[15] and the workgroup writes NaN instead of its outputs; afterwards the counters are out of step: the caller zeroes
 * `sync` before the next launch (samplenet_amd.pointnet.check_chain_errors). */
int sn_fc_chain_forward_supported(int R, int C0, int H, int nl);
int sn_fc_chain_forward(int R, int C0, int H, int nl, const float *a0, const float *const *W, const float *const *bias,
                        const float *const *gamma, const float *const *beta, float *const *running_mean,
                        float *const *running_var, long long *const *num_batches_tracked, const float *eps,
                        const float *momentum, float *const *z, float *const *coef, float *xbuf, unsigned *sync,
                        sn_stream_t stream);
/* sn_fc_chain_forward with the tail of the conv stack in front (samplenet.py:90-101: bn5 / relu / max over the points /
 * fc1..fc3): call sn_conv_stack_forward_bn with pooled = argsel = zsel = NULL -- it then stops after its last GEMM, leaving
 * the last layer's fixed-point sums in acc and the block maxima / minima in pool_val / pool_idx -- and this entry right after:
 * its first stage finalises that BatchNorm (gamma5 .. coef5, clearing acc) and decodes the max-pool from the keys (pooled /
 * argsel / zsel as sn_pool_forward) in EVERY workgroup -- no exchange in front of fc1.  B <= 32 clouds, N % 64 == 0, conv stack
 * of nconv layers ending in C0 = 128 channels, H = 256, nl = 3 (query _supported).  Same results as the two separate calls.
 * pool_val: the keys; pool_idx: unused. */
int sn_fc_chain_forward_pool_supported(int B, int N, int C0, int H, int nl);
int sn_fc_chain_forward_pool(int B, int N, int nconv, long long *acc, const float *pool_val, const int *pool_idx,
                             const float *gamma5, const float *beta5, float *running_mean5, float *running_var5,
                             long long *num_batches_tracked5, float eps5, float momentum5, float *coef5, float *pooled,
                             int *argsel, float *zsel, int H, int nl, const float *const *W, const float *const *bias,
                             const float *const *gamma, const float *const *beta, float *const *running_mean,
                             float *const *running_var, long long *const *num_batches_tracked, const float *eps,
                             const float *momentum, float *const *z, float *const *coef, float *xbuf, unsigned *sync,
                             sn_stream_t stream);
/* sn_fc_chain_forward_pool with the head's OUTPUT layer behind the hidden layers as the chain's last stage: Linear (Co x H) +
 * BatchNorm WITHOUT activation -- the classification task's sampler (classification/models/samplenet_model.py:100-108).  Wo (Co, H),
 * bo (Co), gamma_o / beta_o (Co), running statistics of that BatchNorm (may be NULL), eps_o / momentum_o -> zo (B, Co) pre-BatchNorm
 * output, coef_o (4, Co), y (B, Co) = zo scale + shift.  Co: a multiple of 32, at most H.  One launch instead of this one +
 * sn_layer_forward_bn_out. */
int sn_fc_chain_forward_pool_out_supported(int B, int N, int C0, int H, int nl, int Co);
int sn_fc_chain_forward_pool_out(int B, int N, int nconv, long long *acc, const float *pool_val, const int *pool_idx,
                                 const float *gamma5, const float *beta5, float *running_mean5, float *running_var5,
                                 long long *num_batches_tracked5, float eps5, float momentum5, float *coef5, float *pooled,
                                 int *argsel, float *zsel, int H, int nl, const float *const *W, const float *const *bias,
                                 const float *const *gamma, const float *const *beta, float *const *running_mean,
                                 float *const *running_var, long long *const *num_batches_tracked, const float *eps,
                                 const float *momentum, float *const *z, float *const *coef, float *xbuf, unsigned *sync, int Co,
                                 const float *Wo, const float *bo, const float *gamma_o, const float *beta_o, float *running_mean_o,
                                 float *running_var_o, long long *num_batches_tracked_o, float eps_o, float momentum_o, float *zo,
                                 float *coef_o, float *y, sn_stream_t stream);
/* The FC head's backward (R <= 32 rows) as ONE launch, the mirror of sn_fc_chain_forward.  Stage s = GEMM layer, TOP first:
 * W[s] (Co[s], Ci[s]); below it: zprev[s] (R, Ci[s]) pre-BN output seen through coefprev[s] (4, Ci[s]) (the pooled-feature
 * stage: zsel with the last conv layer's coefficients and bn_rows[s] = B * N; bn_rows < 0: fixed statistics); outputs per
 * stage: dW[s], the layer below's dgamma / dbeta / dbias; db_top: bias gradient of the top layer; aprev[s] / araw[s]: the
 * weight-gradient operand (zprev with ReLU(BN) applied, or a raw input such as the pooled features).  Last stage also
 * returns gout (R, Ci) = the masked gradient and kout (3, Ci) = the dZ coefficients of the BatchNorm below (what
 * sn_conv_stack_backward takes as gsel / kcoef_top).  gy: (R, Co[0]).  Bit-identical to the sn_layer_backward chain.
 * xbuf: ns * 32 * 256 floats of scratch; sync: 16 x 32 unsigned (same layout) of its OWN persistent zero-initialised state (launch epoch, one
 * monotonic arrival counter per stage, epoch-reader count; sync[13] / sync[15] and the NaN poisoning as sn_fc_chain_forward;
 * 16 workgroups must be resident together). */
int sn_fc_chain_backward_supported(int R, int ns, const int *Co, const int *Ci);
int sn_fc_chain_backward(int R, int ns, const int *Co, const int *Ci, const float *gy, const float *const *W,
                         const float *const *zprev, const float *const *coefprev, const long long *bn_rows,
                         float *const *dgamma, float *const *dbeta, float *const *dbias, float *const *dW, float *db_top,
                         const float *const *aprev, const int *araw, float *gout, float *kout, float *xbuf,
                         unsigned *sync, sn_stream_t stream);
/* sn_fc_chain_backward for a head whose output went through a BatchNorm WITHOUT activation (the classification sampler's output
 * layer: sn_layer_forward_bn_out / sn_fc_chain_forward_pool_out): gy is the gradient behind that BatchNorm, zo (R, Co[0]) its
 * input, coef_o (4, Co[0]) the forward's coefficients; the launch opens with that BatchNorm's backward (the arithmetic of
 * sn_bn_output_backward, rows summed in order) and leaves dgamma_o / dbeta_o (Co[0]).  fixed != 0: running statistics. */
int sn_fc_chain_backward_obn(int R, int ns, const int *Co, const int *Ci, const float *gy, const float *zo, const float *coef_o, int fixed,
                             float *dgamma_o, float *dbeta_o, const float *const *W, const float *const *zprev,
                             const float *const *coefprev, const long long *bn_rows, float *const *dgamma, float *const *dbeta,
                             float *const *dbias, float *const *dW, float *db_top, const float *const *aprev, const int *araw, float *gout,
                             float *kout, float *xbuf, unsigned *sync, sn_stream_t stream);
/* sn_bn_finalize for a SHORT matrix z (R, C) (the FC head at batches above 32): statistics in two passes over z itself (mean,
 * then squares around it) instead of from sum / sum-of-squares partials -- behind the max-pool |mean| / std reaches 10..100. */
int sn_bn_batch_stats_twopass(int R, int C, const float *z, const float *gamma, const float *beta, float eps, float momentum,
                              float *running_mean, float *running_var, long long *num_batches_tracked, float *coef,
                              sn_stream_t stream);

/* The BatchNorm WITHOUT activation on the head's output -- the classification task's sampler (classification/models/
 * samplenet_model.py:100-108: fc14b with bn=True, activation_fn=None; the registration sampler has no such layer, registration/
 * src/samplenet.py:59,104).
 *   sn_layer_forward_bn_out   R <= 32 rows, Ci a power of two in 64 .. 512: ONE launch -- z (R, Co) = act(ain) W^T + bias (act =
 *                             relu(scale x + shift) with coef_prev, identity when NULL), training-mode batch statistics over the R
 *                             rows -> coef (4, Co) = scale, shift, mean, invstd + running statistics as torch.nn.BatchNorm1d, and
 *                             y (R, Co) = z scale + shift.  Other shapes: SN_ERR_UNSUPPORTED (sn_linear_forward_rows + the next one).
 *   sn_bn_output_forward      any R, C % 4 == 0: training != 0: two-pass batch statistics of z (as sn_bn_batch_stats_twopass), else
 *                             coefficients from the running statistics; then y = z scale + shift.
 *   sn_bn_output_backward     gy (R, C) -> dz (R, C), dgamma (C), dbeta (C); fixed != 0: the forward ran on running statistics
 *                             (dz = scale gy).  Sums in double, fixed order (deterministic). */
int sn_layer_forward_bn_out(int R, int Ci, int Co, const float *ain, const float *coef_prev, const float *W, const float *bias,
                            float *z, const float *gamma, const float *beta, float eps, float momentum, float *running_mean,
                            float *running_var, long long *num_batches_tracked, float *coef, float *y, sn_stream_t stream);
int sn_bn_output_forward(int R, int C, int training, const float *z, const float *gamma, const float *beta, float eps,
                         float momentum, float *running_mean, float *running_var, long long *num_batches_tracked, float *coef,
                         float *y, sn_stream_t stream);
int sn_bn_output_backward(int R, int C, int fixed, const float *gy, const float *z, const float *coef, float *dz, float *dgamma,
                          float *dbeta, sn_stream_t stream);

/* sn_pool_backward + sn_bn_backward_coef of the last conv layer (R = B * N rows seen by its BatchNorm) in one launch */
int sn_pool_backward_bn(int B, int C, long long R, const float *g, const float *pooled, const float *zsel, float *gsel,
                        const float *coef, float *dgamma, float *dbeta, float *dbias, float *kcoef, sn_stream_t stream);

/* dgrad + wgrad of one layer in ONE launch (both kinds of workgroups resident together); same arguments, no bias column */
int sn_linear_backward(int R, int Ci, int Co, int dz_mode, const float *dy, const float *z, const float *kcoef,
                       const float *gsel, const int *argsel, int npts, const float *W, const float *zprev,
                       const float *coef_prev, float *dyprev, float *stats, float *part, float *dW, sn_stream_t stream);

/* the fused data+weight gradient kernel of a 64/128-channel 1x1 convolution on its own: dYprev, stats partials [G][2][Ci],
 * dW partials [G][Co][Ci], G = sn_linear_wgrad_splits(R,Ci,Co,0); SN_ERR_UNSUPPORTED for other shapes */
int sn_conv_backward_partials(int R, int Ci, int Co, int dz_mode, const float *dy, const float *z, const float *kcoef,
                              const float *gsel, const int *argsel, int npts, const float *W, const float *zprev,
                              const float *coef_prev, float *dyprev, float *stats, float *part, sn_stream_t stream);

/* Task network (registration/models/pcrnet.py:23-41 PointNetFeatures: 1x1 convolutions + ReLU without BatchNorm, max over the
 * points) -- fused form of sn_linear_forward + sn_pool_forward:
 *   sn_linear_forward_maxpool   last layer + max over the npts points of every cloud in one GEMM launch (+ a 2 KB-per-cloud key
 *                               clear and a decode launch): pooled (B,Co) = relu(max_n Z), argsel / zsel (optional) as
 *                               sn_pool_forward; z (R,Co) optional -- NULL: the activations are never written (a branch
 *                               that needs no gradient).  coef_prev: the previous layer's (scale, shift) = (1, 0) table.
 *                               keys: B * 2 * Co 64-bit words of scratch, cleared here unless keys_cleared != 0 (an earlier launch
 *                               on the stream did: sn_pointnet_narrow_forward's zero_keys).  Query _supported (64-aligned shapes). */
int sn_linear_forward_maxpool_supported(int R, int Ci, int Co, int npts);
int sn_linear_forward_maxpool(int R, int Ci, int Co, int npts, const float *ain, const float *coef_prev, const float *W,
                              const float *bias, float *z, unsigned long long *keys, float *pooled, int *argsel, float *zsel,
                              int keys_cleared, sn_stream_t stream);

/*   sn_linear_forward_maxpool_wide   the same for a WIDE last layer (PCRNet: 128 -> 1024): a workgroup keeps the split A fragments
 *                               of its 128 rows in registers for all of its columns, the weights are split once per call into
 *                               wplanes (3 * Co * Ci bf16; planes_ready != 0: already holds the split of this W); the maxima
 *                               leave as one key per column and min(128, npts) rows (scratch: _scratch_bytes) -- no atomics, no
 *                               clear; results bit-identical to sn_linear_forward_maxpool.  Query _supported. */
int sn_linear_forward_maxpool_wide_supported(int R, int Ci, int Co, int npts);
long long sn_linear_forward_maxpool_wide_scratch_bytes(int R, int Ci, int Co, int npts);
int sn_linear_forward_maxpool_wide(int R, int Ci, int Co, int npts, const float *ain, const float *coef_prev, const float *W,
                                   const float *bias, float *z, void *scratch, float *pooled, int *argsel, float *zsel,
                                   void *wplanes, int planes_ready, sn_stream_t stream);
/*   sn_pool_dgrad_sparse        data gradient of that last layer for a BatchNorm-free stack (dZ has ONE non-zero per cloud and
 *                               channel): dyprev (B*npts, Ci) = relu'_prev . sum_c [argsel[b][c] == n] (pooled > 0 ? g : 0)[b][c]
 *                               W[c][:] -- sn_pool_backward + sn_linear_dgrad(DZ_POOL) without the dense (R, Co) operand.
 *                               zprev / coef_prev (scale | shift): the previous layer's pre-activations and operand
 *                               coefficients for its ReLU mask (NULL: no mask).  Deterministic.  Query _supported (npts <= 256). */
int sn_pool_dgrad_sparse_supported(int B, int npts, int Ci, int Co);
int sn_pool_dgrad_sparse(int B, int npts, int Ci, int Co, const float *g, const float *pooled, const int *argsel, const float *W,
                         const float *zprev, const float *coef_prev, float *dyprev, sn_stream_t stream);

/*   sn_pointnet_narrow_forward  the extractor's narrow front 3 -> 64 -> 64 -> 64 -> 128 (ReLU between, no BatchNorm: conv1..conv4 of
 *                               registration/models/pcrnet.py:23-38) in one launch: a wave takes 32 rows through all four layers;
 *                               every layer's pre-activations bit-identical to the layer-by-layer launches.  z1..z3 (R,64): all
 *                               three (a backward will read them) or none; z4 (R,128).  wplanes: 3 * 16384 bf16 for the split
 *                               weights (planes_ready != 0: already holds the split of these weights).  zero_keys / zero_n: optional
 *                               rider -- that many 64-bit words are cleared (the key scratch of the sn_linear_forward_maxpool
 *                               that follows). */
int sn_pointnet_narrow_forward_supported(int R, int c1, int c2, int c3, int c4);
int sn_pointnet_narrow_forward(int R, const float *x, const float *W1, const float *b1, const float *W2, const float *b2, const float *W3,
                               const float *b3, const float *W4, const float *b4, void *wplanes, int planes_ready, float *z1, float *z2,
                               float *z3, float *z4, unsigned long long *zero_keys, int zero_n, sn_stream_t stream);

/*   sn_pointnet_narrow_backward the data gradient back through that front in one launch (frozen weights): dx (R,3) from dz4 (R,128) =
 *                               dL/d(conv4's pre-activations) and the saved z1..z3; wplanes_t: 3 * 16384 bf16 for the transposed split
 *                               weights (planes_ready != 0: already there). */
int sn_pointnet_narrow_backward_supported(int R, int c1, int c2, int c3, int c4);
int sn_pointnet_narrow_backward(int R, const float *dz4, const float *z1, const float *z2, const float *z3, const float *W1, const float *W2,
                                const float *W3, const float *W4, void *wplanes_t, int planes_ready, float *dx, sn_stream_t stream);

/* Linear layers on at most 32 rows (PCRNet's trunk, registration/models/pcrnet.py:56-77): out (R, N) = act((x . [gate > 0]) (R, K) .
 * W^T + bias), the weight stream cut into (32-column tile) x (K slice) workgroups, slices summed in order by the last workgroup to
 * arrive (deterministic); fp32 products as split-bf16 MFMAs.
 *   transposed == 0: W (N, K) -- forward;  != 0: W (K, N) -- data gradient dX = (dY . [y > 0]) W with gate = the layer's output y
 *   gate, bias: optional.  scratch: _scratch_bytes; counters: (N + 31) / 32 zeroed 32-bit words (left zeroed). */
/* sn_pcrnet_head_forward + sn_qrot_forward as ONE launch (main.py:563-571: twist = model(p0, p1); est_transform.rotate(p0)):
 * y (B,7), v (B,N,3) -> twist (B,7), quat (B,4), qnorm (device scalar, may be NULL), out (B,N,3) = v rotated by quat; and the
 * backward of that pair as one launch: grad_out (B,N,3), grad_twist (B,7), grad_quat (B,4), grad_qnorm (device scalar), each
 * may be NULL -> grad_y (B,7) and, when grad_v != NULL, grad_v (B,N,3).  Bit-identical to the two launches each way. */
int sn_pcrnet_head_rot_forward(int B, int N, const float *y, const float *v, float *twist, float *quat, float *qnorm, float *out,
                               sn_stream_t stream);
int sn_pcrnet_head_rot_backward(int B, int N, const float *y, const float *quat, const float *v, const float *grad_out,
                                const float *grad_twist, const float *grad_quat, const float *grad_qnorm, float *grad_v, float *grad_y,
                                sn_stream_t stream);

/* The progressive sampler's nested prefixes (classification/train_samplenet_progressive.py:157-234) as contiguous tensors in ONE launch:
 * src (B, M, C) of 4-byte elements -> dst[j] (B, sizes[j], C) = src[:, :sizes[j], :] (dst: HOST array of nprefix <= 16 device
 * pointers, NULL entries skipped); and the gradient of that in one launch: out (B, M, C) = sum over j of grads[j] zero-padded to
 * M points, ascending j (NULL entries contribute nothing). */
int sn_prefix_pack(int B, int M, int C, int nprefix, const int *sizes, const void *src, void *const *dst, sn_stream_t stream);
int sn_prefix_scatter_sum(int B, int M, int C, int nprefix, const int *sizes, const float *const *grads, float *out, sn_stream_t stream);
/* Clouds of different sizes as one batch of equal-size clouds for a max-pooling extractor without BatchNorm (PCRNet's PointNetFeatures,
 * registration/models/pcrnet.py:23-46): out[(j B + b), m, :] = src[j][b, m mod sizes[j], :] for m < len (src: HOST array of nclouds <= 16
 * device pointers, cloud j (B, sizes[j], C) fp32); the pooled features of every cloud are unchanged, bit for bit.  _backward: the
 * copies' gradients added onto their originals (grads[j] overwritten; NULL entries skipped). */
int sn_cyclic_pad_cat(int B, int len, int C, int nclouds, const int *sizes, const float *const *src, float *out, sn_stream_t stream);
int sn_cyclic_pad_cat_backward(int B, int len, int C, int nclouds, const int *sizes, const float *grad_out, float *const *grads,
                               sn_stream_t stream);
/* Several evaluations of the registration task term against ONE template in one batch (the progressive sampler's prefixes,
 * classification/train_samplenet_progressive.py:181-216 with registration/main.py:557-577 as the task): rows / clouds [e group, (e+1) group)
 * are evaluation e.
 *   sn_pcrnet_head_rot_*_grouped : sn_pcrnet_head_rot_* on R = E group rows, v = the `group` template clouds (row b rotates v[b % group]),
 *                                  qnorm / grad_qnorm one value per evaluation;
 *   sn_chamfer_mean_loss_*_grouped: mean(dist1) + mean(dist2) per evaluation on clouds padded to n1 points by sn_cyclic_pad_cat, nvalid[e]
 *                                  of them real (HOST array): the copies are not counted forward and get / give no gradient backward.
 * Every number equals the evaluation's own ungrouped call on the unpadded cloud, bit for bit. */
/* sn_chamfer_forward for such a padded batch: only the first q_valid[b / q_group] points (device array) of the smaller cloud of pair b
 * are scanned; the copies' dist / idx stay unwritten, everything else equals sn_chamfer_forward's products. */
int sn_chamfer_forward_valid(int B, int m, const float *xyz_small, int n, const float *xyz_large, const int *q_valid, int q_group,
                             float *dist_small, int *idx_small, float *dist_large, int *idx_large, void *workspace,
                             long long workspace_bytes, sn_stream_t stream);
int sn_pcrnet_head_rot_forward_grouped(int R, int N, int group, const float *y, const float *v, float *twist, float *quat, float *qnorm,
                                       float *out, sn_stream_t stream);
int sn_pcrnet_head_rot_backward_grouped(int R, int N, int group, const float *y, const float *quat, const float *v, const float *grad_out,
                                        const float *grad_twist, const float *grad_quat, const float *grad_qnorm, float *grad_y,
                                        sn_stream_t stream);
int sn_chamfer_mean_loss_forward_grouped(int R, int n1, int n2, int group, int nev, const int *nvalid, const float *dist1,
                                         const float *dist2, float *partial, float *loss, sn_stream_t stream);
int sn_chamfer_mean_loss_backward_grouped(int R, int n1, const float *xyz1, int n2, const float *xyz2, int group, int nev, const int *nvalid,
                                          const int *idx1, const int *idx2, const float *grad_loss, float *grad_xyz1, float *grad_xyz2,
                                          sn_stream_t stream);
/* sigma[0] = max(T^2, min_sigma) (registration/src/soft_projection.py:97-99: the projection loss of get_projection_loss) in one launch;
 * its backward is sn_sigma_grad with the upstream gradient as the single partial. */
int sn_sigma_forward(const float *temperature, float min_sigma, float *sigma, sn_stream_t stream);
/* The simplification losses (samplenet.py:171-181) of the first nterms nested prefixes of the simplified cloud, summed ascending
 * (classification/train_samplenet_progressive.py:204-216), behind one node: dq / iq (B,M) the per-query Chamfer products of the full
 * set, d2 / i2 (S,B,N) sn_prefix_point_minima's products; partial 3 * nterms * B floats, argmax1 nterms * B ints (forward -> backward).
 * Bit-identical to nterms sn_simplification_loss_forward / _backward calls on contiguous copies of the prefixes, added ascending. */
int sn_prefix_simplification_loss_forward(int B, int M, int N, int nterms, const int *sizes, const float *weights, const float *dq,
                                          const float *d2, float *partial, int *argmax1, float *loss, sn_stream_t stream);
int sn_prefix_simplification_loss_backward(int B, int M, int N, int nterms, const int *sizes, const float *weights, const float *samp_pc,
                                           const float *ref_pc, const int *iq, const int *i2, const int *argmax1, const float *grad_loss,
                                           float *grad_samp, sn_stream_t stream);
/* Test hook: the auction's level passes in segments (1, default: the other cloud cut into ranges swept by separate workgroups, partial
 * sums added in ascending order by the last to arrive -- thousands of short workgroups instead of 1.56 waves per SIMD at B = 50) or as
 * one range per workgroup (0); returns the previous setting. */
int sn_emd_set_segments(int on);
/* Test hook: sn_emd_loss_fast on the one-sweep form (1, default: every pair's match value evaluated once for cost, grad1 and grad2;
 * 64 x 64 tiles, tile partials added in ascending order) or on the two order-preserving sweeps of sn_emd_loss (0); returns the
 * previous setting.  Both meet the loss bar (cost 1e-5 of the oracle's). */
int sn_emd_set_sweep2d(int on);
int sn_skinny_linear_supported(int R, int K, int N);
long long sn_skinny_linear_scratch_bytes(int R, int K, int N);
int sn_skinny_linear(int R, int K, int N, const float *x, const float *gate, const float *W, int transposed, const float *bias, int relu,
                     float *out, float *scratch, unsigned *counters, sn_stream_t stream);
/* two-part form: input columns k >= ksplit from x2 (R, K - ksplit) with x (R, ksplit) (x2 NULL: x is (R, K)); output columns n >= nsplit
 * to out2 (R, N - nsplit) with out (R, nsplit) (nsplit 0: out is (R, N)); with nsplit > 0 either output may be NULL.  The trunk's
 * first layer reads the two clouds' feature vectors where they lie and its data gradient hands each cloud its own gradient. */
int sn_skinny_linear2(int R, int K, int N, const float *x, const float *x2, int ksplit, const float *gate, const float *W, int transposed,
                      const float *bias, int relu, float *out, float *out2, int nsplit, float *scratch, unsigned *counters,
                      sn_stream_t stream);
/* Weight gradient of such a layer (a trainable trunk: registration/models/pcrnet.py:62-82 under main.py --train-pcrnet):
 * dW (N, K) = (dy . [gate > 0])^T . [x | x2], db (N, optional) = its column sums; R <= 256 rows, fp32 MFMA, rows in ascending
 * order (deterministic).  gate (R, N, optional): the layer's own output (ReLU mask); x2 / ksplit as in sn_skinny_linear2. */
int sn_skinny_wgrad(int R, int K, int N, const float *x, const float *x2, int ksplit, const float *dy, const float *gate, float *dW,
                    float *db, sn_stream_t stream);
/* Test hook: 64-row blocks per workgroup of the xyz layer's statistics pass inside sn_conv_stack_forward_bn (0 = chosen per call:
 * 1 until the batch is large, then up to 8 -- one pair of atomics per channel and workgroup instead of one per block; the integer
 * totals are the same whatever the grouping).  Returns the previous setting. */
int sn_conv_stack_set_in3_blocks(int blocks);
/* sn_linear_forward for a layer of the FC head above 32 rows (rows = clouds, registration/src/samplenet.py:97-104 at a batch of
 * 64 .. ~1000): no statistics (the head takes two-pass statistics from Z, sn_bn_batch_stats_twopass).  While (R / 32) x (Co / 32)
 * workgroups fit the chip once, the R <= 32 kernel runs row block by row block -- every workgroup's operands in flight at once --
 * instead of the 64 x 64 tile kernel's dependent K chunks (512 x 256 -> 256: 8.8 us instead of 14.8); same products, K summed in
 * four partial chains.  Any other shape: sn_linear_forward's dispatch. */
int sn_linear_forward_rows(int R, int Ci, int Co, const float *ain, const float *coef_prev, const float *W, const float *bias, float *z,
                           sn_stream_t stream);

/* sn_emd_loss (samplenet_hip.h) with the reference op's own exponential -- __expf = v_exp_f32(x log2 e), tf_approxmatch_g.cu:52,97,151 --
 * in the auction and in the cost / gradient sweeps: the form a training loss wants (cost within 1e-5 of the oracle); not
 * bit-identical to the three-call composition, whose match matrix keeps the compensated exponential.  Same arguments / scratch. */
int sn_emd_loss_fast(int b, int n, int m, const float *xyz1, const float *xyz2, float *cost, float *grad1, float *grad2,
                     float *temp, sn_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SAMPLENET_HIP_INTERNAL_H */
