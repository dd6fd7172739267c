"""Data parallelism for the sampler: one process per GPU, batch sharded across ranks, ONE gradient
bucket all-reduced per step over RCCL/xGMI (torch.distributed backend "nccl" == RCCL on ROCm).

The reference has no distributed code at all; this is the new multi-GPU layer around the
unmodified module surface.  Design for xGMI (point-to-point, 7 links x ~153 GB/s per GPU):
  * SampleNet has 249,793 fp32 parameters (~1 MB): the exchange is latency-bound, so all gradients
    live in ONE flat buffer (param.grad are views into it) and the whole step costs at most two
    collectives -- no per-parameter collectives, no bucket-size tuning;
  * the FC head's gradients (86 % of the bytes) are complete before the conv stack's backward
    starts: with `overlap=True` they are reduced on a side stream while the conv backward runs,
    the conv gradients follow at the end (two collectives, the first one hidden).  Under a captured
    step (engine.SamplerTrainStep) the same split is made at graph level: graph 1 = forward .. FC
    backward, graph 2 = conv backward, the first collective launched between them on the side stream;
  * BatchNorm uses per-rank batch statistics by default (each replica behaves exactly like the reference at its
    local batch size); syncbn.convert_sync_batchnorm(net, group) switches the head to statistics over all ranks'
    rows (W x B clouds train like one process with W B clouds; a parity mode on the layer-by-layer kernels).
"""
import weakref

import torch
import torch.distributed as dist


class FlatGradAllReducer:
    """Keeps every parameter's .grad as a view into one flat fp32 buffer and averages it across ranks.

    The MLP gradients are WRITTEN into the views by the HIP backward kernels themselves (module._grad_sink, a
    pointnet.GradSink: overwrite on the first backward of a step, accumulate on further ones, views re-bound after
    optimizer.zero_grad()), so there is no per-parameter accumulate / copy kernel; only the temperature goes through
    autograd's accumulate.  Layout: [FC head | temperature | conv stack] -- the first segment is complete before the
    conv stack's backward starts.

    force_collective: issue the collectives even at world size 1 (single-GPU exercise of the RCCL path: tests, bench).
    kernel_written: names of the parameters whose gradients the module's backward writes into the views itself (default: what
    pointnet.param_order names for the HIP model on a GPU, nothing otherwise).
    """

    def __init__(self, module, process_group=None, overlap=True, force_collective=False, kernel_written=None):
        self.module = module
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.force_collective = bool(force_collective and dist.is_initialized())
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
        # the HIP MLP writes its gradients itself; anything else (the CPU model of the gloo test) goes through autograd
        hip_mlp = bool(getattr(module, "use_hip_mlp", False)) and dev.type == "cuda"
        # ... and of the HIP model only the parameters the MLP node differentiates (pointnet.param_order): a BatchNorm behind
        # the last FC layer (classification sampler) is applied by torch on the head's output, its gradients come through
        # autograd like the temperature's and must be zeroed / re-bound with them
        # (kernel_written=: explicit list for modules whose backward fills the sink views itself -- the CPU stand-in of the
        #  multi-process tests)
        if kernel_written is not None:
            kernel_written = set(kernel_written)
        elif hip_mlp:
            from .pointnet import param_order

            kernel_written = set(param_order(module))
        else:
            kernel_written = set()
        sink, off = {}, 0
        self._autograd = []  # (parameter, view): gradients that arrive through autograd's accumulate
        for n, p in order:
            view = self.flat[off:off + p.numel()].view_as(p)
            p.grad = view
            if n in kernel_written:
                sink[n] = view
            else:
                self._autograd.append((p, view))
            off += p.numel()
        if sink:
            from .pointnet import GradSink

            params = dict(named)
            module._grad_sink = GradSink(sink, {n: params[n] for n in sink})
            # (the captured module surface writes this bucket: surface.py.  A WEAK reference: module -> sink -> reducer -> module
            #  would be a cycle that only the garbage collector frees, and a captured graph destroyed by a collection that
            #  happens to run during somebody's stream capture aborts the process)
            module._grad_sink._reducer_ref = weakref.ref(self)
        self.overlap = bool(overlap and self.collective and dev.type == "cuda" and hip_mlp)
        self._side = torch.cuda.Stream(device=dev) if (self.collective and dev.type == "cuda") else None
        self._early_work = None
        self._sentinel = None
        # set by a captured backward whose last node WAS the all-reduce (surface.py): reduce() skips the collective -- unless a
        # contribution that was NOT reduced landed in the bucket in the same step (an op-by-op backward of a second sampled cloud,
        # an irregular upstream): then the collective runs; averaging (reduced + local) over the ranks is reduced + mean(local)
        self._graph_reduced = False
        self._unreduced = False
        self.capture_fork = False  # engine, "graph-fork" mode: the early collective may be issued while a graph is being captured
        if self.overlap:
            module._after_fc_grads = self._early_ready

    @property
    def collective(self):
        return self.world > 1 or self.force_collective

    def disable_overlap(self):
        """No collective from inside backward (a captured step replays backward without Python; the engine splits the
        step into two graphs at the same point instead)."""
        if self._early_work is not None:
            self._early_work.wait()
            self._early_work = None
        self.overlap = False
        if getattr(self.module, "_after_fc_grads", None) is not None:
            self.module._after_fc_grads = None

    def begin_step(self):
        """The next backward overwrites the kernel-written views (no fill needed)."""
        sink = getattr(self.module, "_grad_sink", None)
        if sink is not None:
            sink.reset()
        self._graph_reduced = self._unreduced = False

    def zero_grad(self, keep=None):
        """Start of a step: begin_step() + zero the autograd-accumulated slices.  optimizer.zero_grad() -- either flavour --
        works too (pointnet.GradSink and _rebind() below take the views back); this is just the cheapest form.
        keep: a parameter whose slice a kernel of the step overwrites in place (the engine's temperature sink): not filled."""
        for p, v in self._autograd:
            if p is not keep:
                v.zero_()
        self.begin_step()

    @property
    def autograd_accumulated(self):
        """Number of parameters whose gradients reach the bucket through autograd's accumulate (temperature; a BatchNorm that
        torch applies on the head's output)."""
        return len(self._autograd)

    def _rebind(self, replayed=False):
        """Gradients autograd accumulated into tensors of its own (after zero_grad(set_to_none=True)) move into the bucket.
        replayed: a graph replay has just written EVERY view (also the autograd-accumulated ones: the capture recorded the
        accumulation into the view) -- a dropped .grad is re-bound as it is instead of being taken for "no gradient"."""
        for p, view in self._autograd:
            g = p.grad
            if g is None:
                if not replayed:
                    view.zero_()
            elif g.data_ptr() == view.data_ptr():
                continue
            else:
                view.copy_(g)
            p.grad = view
        # the kernel-written views: after optimizer.zero_grad(set_to_none=True) a graph REPLAY has filled them without any
        # Python running (GradSink.commit re-binds only while the backward executes eagerly or is being captured).
        # zero_grad() drops all of them together: one sentinel decides whether the walk is needed (this runs per step)
        sink = getattr(self.module, "_grad_sink", None)
        if sink:
            if self._sentinel is None:
                n0 = next(iter(sink))
                self._sentinel = (sink.params[n0], sink[n0])
            p0, v0 = self._sentinel
            if p0.grad is None or p0.grad.data_ptr() != v0.data_ptr():
                for n, view in sink.items():
                    p = sink.params[n]
                    if p.grad is None or p.grad.data_ptr() != view.data_ptr():
                        p.grad = view

    # ---------------------------------------------------------------------------------------- collectives
    def _op(self):
        return dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM

    def reduce_early_async(self):
        """All-reduce of the FC-head segment on the side stream, ordered after everything enqueued on the current stream so
        far; reduce() joins it.  (Eager backward calls this through module._after_fc_grads; the captured step between its
        two graphs.)"""
        cur = torch.cuda.current_stream()
        self._side.wait_stream(cur)
        with torch.cuda.stream(self._side):
            self._early_avg = self._avg
            self._early_work = dist.all_reduce(self.flat[: self.n_early], op=self._op(), group=self.group, async_op=True)

    def _early_ready(self):
        if torch.cuda.is_current_stream_capturing() and not self.capture_fork:
            return
        self.reduce_early_async()

    def reduce(self, collective=True, replayed=False):
        """Call after backward(): on return (stream-ordered) every .grad holds the cross-rank mean.
        collective=False: only the .grad bookkeeping (the engine's captured step carries the collectives inside its graph);
        replayed=True: the step was a graph replay (see _rebind)."""
        skip = self._graph_reduced and not self._unreduced  # the module surface's backward graph carried the collective
        self._graph_reduced = self._unreduced = False
        if skip:
            self._rebind(True)
            return
        self._rebind(replayed)
        if not collective or not self.collective:
            return
        # RCCL averages inside the collective (ReduceOp.AVG): no separate scaling kernel; gloo (CPU tests) sums, then scales
        if self._early_work is not None:
            self._all_reduce_mean(self.flat[self.n_early:])
            self._early_work.wait()
            torch.cuda.current_stream().wait_stream(self._side)
            self._early_work = None
            if not self._early_avg and self.world > 1:
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
        if self.world > 1:
            t.mul_(1.0 / self.world)


def shard_batch(x, rank, world):
    """Contiguous equal shard of the global batch for this rank (global batch must divide evenly)."""
    B = x.shape[0]
    if B % world:
        raise ValueError("global batch %d is not divisible by world size %d" % (B, world))
    per = B // world
    return x[rank * per:(rank + 1) * per]
