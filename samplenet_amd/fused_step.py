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
import os

import torch

from . import ops, pointnet

# per-cloud loss reduction inside the backward's first launch (sn_sampler_step_loss_fold): one launch less, but every one of
# the 512 workgroups then re-reduces its cloud's 64 KB of partial keys -- measured +2.2 us per step at B = 32, so OFF
FOLD_LOSS = os.environ.get("SAMPLENET_AMD_FOLD_LOSS", "0") == "1"
# per-point minima combined across a cloud's scan workgroups by atomicMax on inverted keys: no reduction launch between the
# scan and the backward (sn_pairscan_forward_keys / sn_sampler_step_loss_keys)
KEYS_LOSS = os.environ.get("SAMPLENET_AMD_KEYS_LOSS", "1") != "0"
# sigma-gradient / loss-value launch on a side stream beside the MLP backward (it is off the critical path): inside the
# captured graph the fork / join costs far more than the 5 us launch it hides (measured +36 us per step) -> OFF
TAIL_STREAM = os.environ.get("SAMPLENET_AMD_TAIL_STREAM", "0") == "1"
# the loss side's scalar tail (sigma gradient, loss value, key-table reset) rides in the closing kernel of the conv stack's
# backward instead of a launch of its own
DEFER_TAIL = os.environ.get("SAMPLENET_AMD_DEFER_TAIL", "1") != "0"


class SamplerStepFunction(torch.autograd.Function):
    """loss, simp (B,3,M), proj (B,M,3) = step(net, x (B,N,3)); differentiable w.r.t. the head's parameters and the
    projection temperature (simp / proj are returned detached: the loss is the only differentiable output)."""

    @staticmethod
    def forward(ctx, net, x_bnc, temperature, K, min_sigma, alpha, lmbda, weight, t_sink, defer_value, *params):
        ops._need_gpu(x_bnc, temperature)
        x = ops._f32c(x_bnc)
        B = x.shape[0]
        M = net.num_out_points
        with torch.cuda.device(x.device):
            _, saved = pointnet.forward_impl(net, x, True, skip_last=True)
            fc4 = net.fc4
            y = torch.empty(B, 3, M, device=x.device, dtype=torch.float32)
            fc = (saved["zf"][2], saved["cf"][2], fc4.weight.detach(), fc4.bias.detach())
            keys = None
            if defer_value and KEYS_LOSS and x.shape[1] <= 2048:
                # persistent zeroed table of inverted per-point keys (every step leaves it zero again)
                keys = getattr(net, "_colmin_keys", None)
                if keys is None or keys.device != x.device or keys.numel() != B * x.shape[1]:
                    keys = torch.zeros(B * x.shape[1], device=x.device, dtype=torch.int64)
                    net._colmin_keys = keys
                elif getattr(net, "_colmin_keys_pending", False):
                    keys.zero_()  # a forward whose backward never ran left its minima behind
                net._colmin_keys_pending = True
            loss, proj, state = ops.step_loss_forward(x, y, fc, temperature, K, min_sigma, alpha, lmbda, weight, defer_value,
                                                      fold=bool(defer_value) and FOLD_LOSS and x.shape[1] <= 2048, keys=keys)
        ctx.net, ctx.saved, ctx.state = net, saved, state
        ctx.x, ctx.y, ctx.temperature = x, y, temperature
        ctx.cfg = (K, float(min_sigma), float(alpha), float(lmbda), float(weight))
        ctx.t_sink = t_sink
        ctx.mark_non_differentiable(y, proj)
        ctx.set_materialize_grads(False)
        return loss[0], y, proj

    @staticmethod
    def backward(ctx, grad_loss, _gy=None, _gproj=None):
        nparams = len(pointnet.PARAM_ORDER)
        if grad_loss is None:
            return (None,) * (10 + nparams)
        net = ctx.net
        sink = getattr(net, "_grad_sink", None)
        if len(ctx.state) > 5 and ctx.state[5][0] == "keys":
            if getattr(ctx, "keys_consumed", False):
                raise RuntimeError("SamplerStepFunction: the key table of this forward was already consumed by a backward "
                                   "(set SAMPLENET_AMD_KEYS_LOSS=0 to backpropagate the same step twice)")
            ctx.keys_consumed = True
        with torch.cuda.device(ctx.y.device):
            keys_mode = len(ctx.state) > 5 and ctx.state[5][0] == "keys"
            # keys mode: the sigma-gradient / loss-value / key-reset launch runs on a side stream beside the MLP backward
            tail = ops.tail_stream(ctx.y.device) if (keys_mode and TAIL_STREAM) else None
            # ... or, better, inside the closing kernel of the conv stack's backward: no launch of its own at all
            blob = None
            if keys_mode and DEFER_TAIL and tail is None and pointnet.conv_stack_backward_supported(net, ctx.x.shape[0], ctx.x.shape[1]):
                import ctypes

                blob = ctypes.create_string_buffer(ops.lib.sn_step_tail_bytes())
            res = ops.step_loss_backward(ctx.x, ctx.y, ctx.temperature, ctx.state, ctx.cfg, grad_loss, ctx.t_sink, tail, blob)
            gQ, gT = res[0], res[1]
            net._colmin_keys_pending = False  # (the backward's last launch re-zeroed the key table)
            grads = pointnet.backward_impl(net, ctx.saved, gQ.view(gQ.shape[0], -1), sink, getattr(net, "_after_fc_grads", None),
                                           step_tail=blob)
            if tail is not None:
                torch.cuda.current_stream().wait_stream(tail)  # join (also what ends the fork inside a graph capture)
            del res  # (scratch of the side-stream launch: released only behind the join)
        g_temp = None
        if ctx.t_sink is None and ctx.needs_input_grad[2]:
            g_temp = gT.reshape(ctx.temperature.shape)
        return (None, None, g_temp) + (None,) * 7 + tuple(
            None if (sink is not None and n in sink) else grads[n] for n in pointnet.PARAM_ORDER)


def sampler_step(net, x_bnc, alpha, lmbda, weight, t_sink=None, defer_value=False):
    """-> (loss, simp (B,3,M), proj (B,M,3)) for a training-mode SampleNet with projection on a (B,N,3) batch."""
    sd = dict(net.named_parameters())
    params = [sd[n] for n in pointnet.PARAM_ORDER]
    proj = net.project
    return SamplerStepFunction.apply(net, x_bnc, proj._temperature, proj._group_size, proj._min_sigma_f, alpha, lmbda, weight,
                                     t_sink, defer_value, *params)
