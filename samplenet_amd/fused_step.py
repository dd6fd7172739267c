"""The sampler's whole training step -- PointNet head, soft projection, loss -- behind ONE autograd node.

    y    = head(x)                          samplenet.py:90-104   (conv/bn/relu x5, max-pool, fc/bn/relu x3, fc4)
    proj = SoftProjection(x, y)             soft_projection.py:138-152
    L    = alpha * L_simp(x, y) + lmbda * sigma + mean(proj)        main.py:507-531 with the benchmark's stand-in task term

Why one node: the head's last layer (fc4: 32 x 256 -> 192 at the benchmark shape) is a launch of its own that only
produces the 64 query points per cloud the pair scan then reads back.  Here the wave that scans query j computes its three
coordinates itself (sn_pairscan_forward_partial_fc), so fc4's forward launch disappears; its backward is unchanged
(pointnet.backward_impl receives d L / d y from the loss kernels).  Used by engine.SamplerTrainStep's fast path; the
op-by-op modules (SampleNet.forward + get_simplification_loss + get_projection_loss) compute the same step.
"""
import weakref

import torch

from . import ops, pointnet

# Test hooks (no environment switches; the defaults are the product path).  KEYS_LOSS: per-point minima combined across a
# cloud's scan workgroups by atomicMax on inverted keys -- no reduction launch between the scan and the backward
# (sn_pairscan_forward_keys / sn_sampler_step_loss_keys); False = the partial-key route every N > 2048 batch takes.
# DEFER_TAIL: the loss side's scalar tail (sigma gradient, loss value, key-table reset) rides in the closing kernel of the conv
# stack's backward instead of a launch of its own; False = the route unsupported conv-stack shapes take.
KEYS_LOSS = True
DEFER_TAIL = True


class _KeyToken:
    """Identity of the forward that currently owns a module's persistent key table (held by its ctx)."""

    __slots__ = ("__weakref__",)


class SamplerStepFunction(torch.autograd.Function):
    """loss, simp (B,3,M), proj (B,M,3) = step(net, x (B,N,3)); differentiable w.r.t. the head's parameters and the
    projection temperature.
      mean_proj=True : loss = alpha * L_simp + lmbda * sigma + mean(proj) -- the benchmark's stand-in task term inside the node;
                       simp / proj are returned detached, the loss is the only differentiable output.
      mean_proj=False: loss = alpha * L_simp + lmbda * sigma, and proj is a DIFFERENTIABLE output: a task loss built on it
                       outside the node (registration/main.py:507-531: the task network sits on the projected points) sends
                       its gradient back in, and the loss backward takes it as an explicit upstream tensor
                       (sn_sampler_step_loss_keys(grad_proj)) -- same scan, same fc4-in-scan, same deferred tail.  Needs the
                       keys-mode step (defer_value, N <= 2048); the loss VALUE is valid once the backward ran."""

    @staticmethod
    def forward(ctx, net, x_bnc, temperature, K, min_sigma, alpha, lmbda, weight, t_sink, defer_value, mean_proj, *params):
        ops._need_gpu(x_bnc, temperature)
        x = ops._f32c(x_bnc)
        B = x.shape[0]
        M = net.num_out_points
        with torch.cuda.device(x.device):
            yo, saved = pointnet.forward_impl(net, x, True, skip_last=True)
            if yo is None:  # the wave that scans query j computes its coordinates itself (the head's output layer inside the scan)
                fc4 = getattr(net, "fc%d" % getattr(net, "num_fc_layers", 4))
                y = torch.empty(B, 3, M, device=x.device, dtype=torch.float32)
                fc = (saved["zf"][-1], saved["cf"][-1], fc4.weight.detach(), fc4.bias.detach())
            else:
                # classification sampler: a BatchNorm over the clouds sits on the head's output (pointnet.out_bn) -- every cloud's
                # queries exist before any scan starts: the scan reads them (one more launch forward, one more backward)
                y, fc = yo.view(B, 3, M), None
            keys = None
            token = None
            if defer_value and KEYS_LOSS and x.shape[1] <= 2048:
                # persistent zeroed table of inverted per-point keys (every step's backward leaves it zero again).  It belongs
                # to ONE forward at a time: a second forward before the owner's backward (two sampler passes under one loss,
                # main.py:516-524) gets a private table instead of clobbering the owner's minima; an owner that was dropped
                # without a backward is detected through its dead token and the table is cleaned.
                keys = getattr(net, "_colmin_keys", None)
                if keys is None or keys.device != x.device or keys.numel() != B * x.shape[1]:
                    keys = torch.zeros(B * x.shape[1], device=x.device, dtype=torch.int64)
                    net._colmin_keys, net._colmin_keys_owner = keys, None
                owner = getattr(net, "_colmin_keys_owner", None)
                if owner is not None and owner() is None:
                    keys.zero_()  # the forward that owned the table never ran its backward
                    owner = None
                token = _KeyToken()
                if owner is None:
                    net._colmin_keys_owner = weakref.ref(token)
                else:
                    keys = torch.zeros(B * x.shape[1], device=x.device, dtype=torch.int64)
            loss, proj, state = ops.step_loss_forward(x, y, fc, temperature, K, min_sigma, alpha, lmbda, weight, defer_value,
                                                      keys=keys)
        ctx.net, ctx.saved, ctx.state = net, saved, state
        ctx.x, ctx.y, ctx.temperature = x, y, temperature
        ctx.cfg = (K, float(min_sigma), float(alpha), float(lmbda), float(weight))
        ctx.t_sink = t_sink
        ctx.keys_token = token
        ctx.mean_proj = bool(mean_proj)
        if not mean_proj and keys is None:
            raise RuntimeError("SamplerStepFunction: an outside task loss (mean_proj=False) needs the keys-mode step "
                               "(defer_value=True, N <= 2048)")
        if mean_proj:
            ctx.mark_non_differentiable(y, proj)
        else:
            ctx.mark_non_differentiable(y)
        ctx.set_materialize_grads(False)
        return loss[0], y, proj

    @staticmethod
    def backward(ctx, grad_loss, _gy=None, gproj=None):
        nparams = len(pointnet.param_order(ctx.net))
        if grad_loss is None and (ctx.mean_proj or gproj is None):
            return (None,) * (11 + nparams)
        net = ctx.net
        if ctx.mean_proj:
            gproj = None
        else:  # the loss here has no mean(proj) term: an absent upstream gradient is a zero one, not the implicit constant
            B, _, M = ctx.y.shape
            gproj = torch.zeros(B, M, 3, device=ctx.y.device) if gproj is None else ops._f32c(gproj)
            if grad_loss is None:
                grad_loss = torch.zeros((), device=ctx.y.device)
        sink, owner = pointnet.sink_for_backward(net)
        if len(ctx.state) > 5 and ctx.state[5][0] == "keys":
            if getattr(ctx, "keys_consumed", False):
                raise RuntimeError("SamplerStepFunction: the key table of this forward was already consumed by a backward "
                                   "(call the step again instead of backpropagating one forward twice)")
            ctx.keys_consumed = True
        with torch.cuda.device(ctx.y.device):
            keys_mode = len(ctx.state) > 5 and ctx.state[5][0] == "keys"
            # keys mode: the sigma-gradient / loss-value / key-reset work rides in the closing kernel of the conv stack's backward
            blob = None
            if keys_mode and DEFER_TAIL and pointnet.conv_stack_backward_supported(net, ctx.x.shape[0], ctx.x.shape[1]):
                import ctypes

                blob = ctypes.create_string_buffer(ops.lib.sn_step_tail_bytes())
            res = ops.step_loss_backward(ctx.x, ctx.y, ctx.temperature, ctx.state, ctx.cfg, grad_loss, ctx.t_sink, blob, gproj)
            gQ, gT = res[0], res[1]
            if keys_mode:  # (the backward's last launch re-zeroed the key table: hand the persistent one back)
                kown = getattr(net, "_colmin_keys_owner", None)
                if kown is not None and kown() is ctx.keys_token:
                    net._colmin_keys_owner = None
            grads = pointnet.backward_impl(net, ctx.saved, gQ.view(gQ.shape[0], -1), sink,
                                           getattr(net, "_after_fc_grads", None) if sink is not None else None, step_tail=blob)
            if owner is not None:
                owner.commit(None if sink is not None else grads)
            del res, gproj  # (scratch the deferred tail reads: released only behind the conv backward)
        g_temp = None
        if ctx.t_sink is None and ctx.needs_input_grad[2]:
            g_temp = gT.reshape(ctx.temperature.shape)
        return (None, None, g_temp) + (None,) * 8 + tuple(
            None if (owner is not None and n in owner) else grads[n] for n in pointnet.param_order(net))


def sampler_step(net, x_bnc, alpha, lmbda, weight, t_sink=None, defer_value=False, mean_proj=True):
    """-> (loss, simp (B,3,M), proj (B,M,3)) for a training-mode SampleNet with projection on a (B,N,3) batch.
    mean_proj=False: proj is differentiable and the loss carries no task term (see SamplerStepFunction)."""
    if net.__dict__.get("_sn_sync_bn") is not None:
        raise RuntimeError("sampler_step runs on per-process BatchNorm statistics; a module converted by "
                           "syncbn.convert_sync_batchnorm takes the layer-by-layer route (SampleNet.forward / engine.SamplerTrainStep)")
    params = pointnet.param_list(net)
    proj = net.project
    return SamplerStepFunction.apply(net, x_bnc, proj._temperature, proj._group_size, proj._min_sigma_f, alpha, lmbda, weight,
                                     t_sink, defer_value, mean_proj, *params)


def external_task_supported(net, x_bnc):
    """True when sampler_step(..., defer_value=True, mean_proj=False) can run this batch (keys-mode scan: N <= 2048 and a batch
    small enough that clouds are split over workgroups)."""
    B, N, _ = x_bnc.shape
    return bool(KEYS_LOSS and N <= 2048 and ops.lib.sn_pairscan_colmin_splits(B, N, net.num_out_points) > 1)


class _DirectCtx:
    """Stand-in for the autograd context when the engine runs the node's forward / backward itself (split capture)."""

    needs_input_grad = (False,) * 64

    def mark_non_differentiable(self, *a):
        pass

    def set_materialize_grads(self, v):
        pass


def sampler_step_direct(net, x_bnc, alpha, lmbda, weight, t_sink, grad_loss, after_fc):
    """forward + backward of SamplerStepFunction without autograd, on the calling thread; every gradient must have a sink
    (FlatGradAllReducer bucket).  after_fc() is invoked once the FC head's gradients have been enqueued."""
    if net.__dict__.get("_sn_sync_bn") is not None:
        raise RuntimeError("sampler_step_direct runs on per-process BatchNorm statistics (see sampler_step)")
    sink = getattr(net, "_grad_sink", None)
    if sink is None or (net.project._temperature.requires_grad and t_sink is None):
        raise RuntimeError("sampler_step_direct needs a gradient sink for every parameter (FlatGradAllReducer)")
    params = pointnet.param_list(net)
    proj = net.project
    ctx = _DirectCtx()
    prev = getattr(net, "_after_fc_grads", None)
    net._after_fc_grads = after_fc
    try:
        with torch.no_grad():
            loss, y, p = SamplerStepFunction.forward(ctx, net, x_bnc, proj._temperature, proj._group_size, proj._min_sigma_f, alpha,
                                                     lmbda, weight, t_sink, True, True, *params)
            SamplerStepFunction.backward(ctx, grad_loss)
    finally:
        net._after_fc_grads = prev
    return loss, y, p
