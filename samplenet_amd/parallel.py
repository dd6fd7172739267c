"""Data parallelism for the sampler: one process per GPU, batch sharded across ranks, ONE gradient
bucket all-reduced per step over RCCL/xGMI (torch.distributed backend "nccl" == RCCL on ROCm).

The reference has no distributed code at all; this is the new multi-GPU layer around the
unmodified module surface.  Design for xGMI (point-to-point, 7 links x ~153 GB/s per GPU):
  * SampleNet has 249,793 fp32 parameters (~1 MB): the exchange is latency-bound, so all gradients
    live in ONE flat buffer (param.grad are views into it) and the whole step costs exactly one
    all-reduce -- no per-parameter collectives, no bucket-size tuning;
  * the FC head's gradients (86 % of the bytes) are complete before the conv stack's backward
    starts: with `overlap=True` they are reduced on a side stream while the conv backward runs,
    the conv gradients follow at the end (two collectives, the first one hidden).
  * BatchNorm uses per-rank batch statistics (each replica behaves exactly like the reference at its
    local batch size); SyncBatchNorm is not applied.
"""
import torch
import torch.distributed as dist


class FlatGradAllReducer:
    """Keeps every parameter's .grad as a view into one flat fp32 buffer and averages it across ranks.

    The MLP gradients are WRITTEN into the views by the HIP backward kernels themselves (module._grad_sink), so
    there is no per-parameter accumulate / copy kernel; only the temperature goes through autograd's accumulate
    and is zeroed by zero_grad().  Layout: [FC head | temperature | conv stack] -- the first segment is complete
    before the conv stack's backward starts.
    """

    def __init__(self, module, process_group=None, overlap=True):
        self.module = module
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self._avg = dist.is_initialized() and dist.get_backend(process_group) == "nccl"
        self._early_avg = False
        named = [(n, p) for n, p in module.named_parameters() if p.requires_grad]
        early = [(n, p) for n, p in named if n.startswith(("fc", "bn_fc"))]
        other = [(n, p) for n, p in named if n.startswith("project")]
        late = [(n, p) for n, p in named if (n, p) not in early and (n, p) not in other]
        order = early + other + late
        self.n_early = sum(p.numel() for _, p in early + other)
        total = sum(p.numel() for _, p in order)
        dev = order[0][1].device
        self.flat = torch.zeros(total, device=dev, dtype=torch.float32)
        sink, off = {}, 0
        self._autograd_slices = []
        for n, p in order:
            view = self.flat[off:off + p.numel()].view_as(p)
            p.grad = view
            if n.startswith("project"):
                self._autograd_slices.append(view)
            else:
                sink[n] = view
            off += p.numel()
        if getattr(module, "use_hip_mlp", False):
            module._grad_sink = sink
        else:  # torch MLP: every gradient arrives through autograd's accumulate
            self._autograd_slices = [self.flat]
        self.overlap = bool(overlap and self.world > 1 and dev.type == "cuda" and getattr(module, "use_hip_mlp", False))
        self._side = torch.cuda.Stream(device=dev) if self.overlap else None
        self._early_work = None
        if self.overlap:
            module._after_fc_grads = self._early_ready

    def disable_overlap(self):
        """Single collective per step (what a captured whole-step graph needs: the early all-reduce is issued from inside
        backward, which the graph replays without Python)."""
        if self._early_work is not None:
            self._early_work.wait()
            self._early_work = None
        self.overlap = False
        if getattr(self.module, "_after_fc_grads", None) is not None:
            self.module._after_fc_grads = None

    def zero_grad(self):
        for v in self._autograd_slices:
            v.zero_()

    def _early_ready(self):
        if torch.cuda.is_current_stream_capturing():
            return
        cur = torch.cuda.current_stream()
        self._side.wait_stream(cur)
        with torch.cuda.stream(self._side):
            self._early_avg = self._avg
            self._early_work = dist.all_reduce(self.flat[: self.n_early], op=dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM,
                                               group=self.group, async_op=True)

    def reduce(self):
        """Call after backward(): on return (stream-ordered) every .grad holds the cross-rank mean."""
        if self.world == 1:
            return
        # RCCL averages inside the collective (ReduceOp.AVG): no separate scaling kernel; gloo (CPU tests) sums, then scales
        if self._early_work is not None:
            self._all_reduce_mean(self.flat[self.n_early:])
            self._early_work.wait()
            torch.cuda.current_stream().wait_stream(self._side)
            self._early_work = None
            if not self._early_avg:
                self.flat[: self.n_early].mul_(1.0 / self.world)
        else:
            self._all_reduce_mean(self.flat)

    def _all_reduce_mean(self, t):
        if self._avg:
            try:
                dist.all_reduce(t, op=dist.ReduceOp.AVG, group=self.group)
                return
            except (RuntimeError, ValueError):  # a backend build without AVG: raised before anything is enqueued
                self._avg = False
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        t.mul_(1.0 / self.world)


def shard_batch(x, rank, world):
    """Contiguous equal shard of the global batch for this rank (global batch must divide evenly)."""
    B = x.shape[0]
    if B % world:
        raise ValueError("global batch %d is not divisible by world size %d" % (B, world))
    per = B // world
    return x[rank * per:(rank + 1) * per]
