"""Captured work behind the drop-in module surface.

An unmodified training script (registration/main.py:507-531, 346-351) drives the sampler like this:

    simp, proj = sampler(x)
    l_simp = sampler.get_simplification_loss(x, simp, M, gamma, delta)
    l_proj = sampler.get_projection_loss()
    loss   = task(proj) + ALPHA * l_simp + LMBDA * l_proj
    optimizer.zero_grad(); loss.backward(); optimizer.step()

Issued op by op that is ~70 kernel launches behind ~0.8 ms of Python for 0.18 ms of GPU work.  Here the same four calls run
on TWO hipGraphs per configuration (batch shape, device, parameter storage):

    forward graph   conv stack (6 launches) -> FC chain (1) -> keys-mode pair scan with fc4 inside (1) -> values of L_simp and
                    sigma + the (B,M,3) copy of the simplified cloud (2)          [the launches of fused_step.SamplerStepFunction]
    backward graph  Chamfer + soft-projection backward (1) -> FC chain backward (1) -> conv stack backward with the step
                    tail (5); gradients land in one flat bucket whose views become the parameters' .grad

behind ONE autograd node with four differentiable outputs (simp, proj, L_simp, sigma): the two loss getters hand out the
node's own outputs, so whatever the script does with them (weights, sums, a task network on proj) reaches the node's backward
as three upstream gradients, which one small launch gathers into the static operands of the backward graph.

A configuration is captured after WARM_STEPS eager steps.  Everything irregular takes the eager route on the same data and
stays correct: a second forward before the first one's backward (main.py:516-524) runs op by op, an upstream gradient on
`simp` itself or a loss weight other than the captured one runs the backward launches eagerly on the graph's activations,
a changed parameter storage / device / dtype drops the graphs (SampleNet._apply; pointer signature checked per step).

Outputs: the graphs write simp / proj / the loss values into ONE static block; what the script receives is a copy of that
block (one small launch), so tensors kept across steps (logging lists, an EMA of proj) keep their values as with the
reference module.  `net.surface_static_outputs = True` hands out the static block itself (the next training forward of the
same configuration then overwrites it in place, as torch.cuda.make_graphed_callables does).

Gradients are written straight into .grad (views of the plan's bucket; torch semantics kept: overwrite when .grad is None,
accumulate otherwise) -- no AccumulateGrad nodes run, so per-parameter autograd hooks would not fire.  Therefore the surface
steps aside, op by op, with ONE warning, whenever something may depend on them:
  * a parameter carries a hook (Tensor.register_hook, register_post_accumulate_grad_hook, optimizer-in-backward);
  * torch.distributed runs with more than one rank and no samplenet_amd.parallel.FlatGradAllReducer is attached to the
    module -- torch DistributedDataParallel hangs its reducer on the AccumulateGrad nodes, which a child module cannot see.
    (`net.graph_surface = "force"` overrides this for scripts that all-reduce p.grad themselves after backward().)
Data parallelism ON the captured surface: attach a FlatGradAllReducer -- the plan's gradient bucket IS the reducer's flat
bucket (the backward graph writes the reducer's views), and where the backend can be captured (RCCL) the all-reduce is the
last node of the backward graph; `reducer.reduce()` after `backward()` then only re-binds .grad (with gloo it runs the
collective as usual).
"""
import ctypes
import operator
import warnings
import weakref

import torch
import torch.distributed as _dist

from . import ops, pointnet
from ._lib import check, lib, ptr, stream_of

ENABLED = True   # test hook / global switch (per module: net.graph_surface)
WARM_STEPS = 2   # eager steps of a configuration before its graphs are captured
MAX_PLANS = 3    # pairs of graphs per configuration (forwards that may wait for their backward at the same time)
MAX_CONFIGS = 4  # configurations (batch shape, device) of one module that keep graphs: the least recently used one gives its up
_suspend = 0


_VIEW_OWNER = {}  # id(gradient view) -> (weak reference to it, weak reference to its plan); entries leave with their view


def _register_view(v, plan):
    key = id(v)
    _VIEW_OWNER[key] = (weakref.ref(v, lambda _r, k=key: _VIEW_OWNER.pop(k, None)), weakref.ref(plan))


class suspended:
    """with surface.suspended(): the module surface runs op by op (callers that place the launches themselves: the engine)."""

    def __enter__(self):
        global _suspend
        _suspend += 1

    def __exit__(self, *exc):
        global _suspend
        _suspend -= 1


class _NoGuard:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NO_GUARD = _NoGuard()


def _device_guard(dev):
    """torch.cuda.device(dev), skipped when dev is the current device already (the context manager costs ~4 us per use)."""
    return _NO_GUARD if torch.cuda.current_device() == dev.index else torch.cuda.device(dev)


class _Token:
    __slots__ = ("__weakref__",)


# A hipGraph destroyed while a stream of the calling thread is capturing raises hipErrorStreamCaptureUnsupported from the
# destructor (~CUDAGraph) and ABORTS the process -- and a plan can lose its last reference at any time: a pruned record, a dropped
# module, a garbage collection (round 5: bench.py's configs[4] leg died exactly so, inside its hand capture).  So the owners of
# captured graphs (surface._Plan, graphed._Plan, engine.SamplerTrainStep) never let their graphs die with them: __del__ hands the
# torch.cuda.CUDAGraph objects to this list, which is emptied at safe points only (a module-surface / graphed call or a plan
# build of a thread that is NOT capturing).
_GRAVE = []


def bury(*graphs):
    try:
        _GRAVE.extend(g for g in graphs if g is not None)
    except Exception:  # noqa: BLE001 -- interpreter shutdown
        pass


def flush_grave():
    if _GRAVE and not torch.cuda.is_current_stream_capturing():
        dead = _GRAVE[:]
        del _GRAVE[:]
        del dead  # (the graphs' destructors run here)


def _collect_before_capture():
    """Garbage that holds captured graphs (a dropped module's plans inside a reference cycle) is collected BEFORE a capture
    begins, and the buried graphs go now."""
    import gc

    gc.collect()
    flush_grave()


class _Live:
    """What the loss getters need to recognise the tensors of a captured forward that still waits for its backward.  The record of
    the LATEST forward holds its outputs strongly (the getters are called right behind it); when another forward follows, the
    older records keep weak references only -- a forward whose outputs the script dropped must not stay "in flight" because of
    this bookkeeping (its plan is free again as soon as the outputs are gone)."""

    __slots__ = ("x", "x_version", "_simp", "_lsimp", "_sigma", "plan", "t_version", "token", "weak")

    def demote(self):
        if not self.weak:
            self._simp, self._lsimp, self._sigma = weakref.ref(self._simp), weakref.ref(self._lsimp), weakref.ref(self._sigma)
            self.x = weakref.ref(self.x)
            self.weak = True

    @property
    def simp(self):
        return self._simp() if self.weak else self._simp

    @property
    def lsimp(self):
        return self._lsimp() if self.weak else self._lsimp

    @property
    def sigma(self):
        return self._sigma() if self.weak else self._sigma

    @property
    def cloud(self):
        return self.x() if self.weak else self.x


_data_ptr = torch.Tensor.data_ptr
_is = operator.is_


def _hyper(net):
    """Everything a plan's graphs have baked in besides tensors: the projection's hyper-parameters, the output size, the
    train / eval flag of every sub-module, the gradient sink (a FlatGradAllReducer's) the backward graph writes."""
    pr = net.project
    return (net.num_out_points, pr._group_size, pr._min_sigma_f, pr._temperature_floor, tuple([m.training for m in net._modules.values()]),
            id(net.__dict__.get("_grad_sink")), bool(net.__dict__.get("surface_static_outputs", False)),
            bool(net.__dict__.get("_sn_surface_simp_grad", False)))


class _Guard:
    """What a plan's graphs have baked in, in a form that is cheap to re-check per step (~10 us): the sub-modules are still
    the module's, every parameter / buffer is still the object at the address the graphs read and write, every parameter
    still wants its gradient, the BatchNorm constants are unchanged."""

    def __init__(self, net):
        self.mods, self.tens, self.consts = [], [], []
        self.net_ref = weakref.ref(net)
        self.hyper = _hyper(net)
        for name, m in net._modules.items():
            self.mods.append((net._modules, name, m))
            for d in (m._parameters, m._buffers):
                for k, t in d.items():
                    if t is not None:
                        self.tens.append((d, k, t, t.data_ptr(), t.requires_grad))
            if isinstance(m, torch.nn.BatchNorm1d):
                self.consts.append((m, m.eps, m.momentum, m.track_running_stats))
        self._slots = [(d, k) for d, k, _, _, _ in self.tens]
        self._objs = [t for _, _, t, _, _ in self.tens]
        self._ptrs = [p for _, _, _, p, _ in self.tens]
        self._rg = [r for _, _, _, _, r in self.tens]

    def ok(self):
        net = self.net_ref()
        if net is None or _hyper(net) != self.hyper:
            return False
        for d, k, m in self.mods:
            if d.get(k) is not m:
                return False
        # (batched: list comprehensions / map instead of a Python-level loop with three tests per tensor: ~15 -> ~7 us)
        if not all(map(_is, [d.get(k) for d, k in self._slots], self._objs)):
            return False
        if list(map(_data_ptr, self._objs)) != self._ptrs or [t.requires_grad for t in self._objs] != self._rg:
            return False
        for m, eps, mom, trs in self.consts:
            if m.eps != eps or m.momentum != mom or m.track_running_stats != trs:
                return False
        return True


class _Plan:
    def __init__(self, net, x, weight):
        self.net_ref = weakref.ref(net)
        self.weight = float(weight)
        self.owner = None
        B, N, _ = x.shape
        M, K = net.num_out_points, net.project._group_size
        self.shape = (B, N, M, K)
        dev = x.device
        self.dev = dev
        self.min_sigma = net.project._min_sigma_f
        params = pointnet.param_list(net)
        names = pointnet.param_order(net)
        T = net.project._temperature
        total = sum(p.numel() for p in params)
        self.sink = net.__dict__.get("_grad_sink")
        red = getattr(self.sink, "reducer", None) if self.sink is not None else None
        self._reducer_ref = weakref.ref(red) if red is not None else None  # (weak: reducer -> module -> plans would be a cycle)
        self.static_out = bool(net.__dict__.get("surface_static_outputs", False))
        # a loss that hangs off the simplified cloud itself (the progressive sampler's prefix losses, a script's own Chamfer on
        # simp): its gradient enters the backward graph as one more static operand, added to the loss kernel's dL/dQ
        self.with_simp = bool(net.__dict__.get("_sn_surface_simp_grad", False))
        self.up_simp_dirty = False
        with torch.cuda.device(dev):
            self.x = torch.empty(B, N, 3, device=dev, dtype=torch.float32)
            self.up_scalars = torch.zeros(2, device=dev, dtype=torch.float32)
            self.up_proj = torch.zeros(B, M, 3, device=dev, dtype=torch.float32)
            self.up_simp = torch.zeros(B, M, 3, device=dev, dtype=torch.float32) if self.with_simp else None
            self.keys = torch.zeros(B * N, device=dev, dtype=torch.int64)
            # one block for everything the script receives: [simp (B,M,3) | proj (B,M,3) | values (8)], sections 256-byte aligned
            sec = (B * M * 3 + 63) // 64 * 64
            self.outbuf = torch.zeros(2 * sec + 64, device=dev, dtype=torch.float32)
            self.out_sec = sec
            self.views, self.view_list, off = {}, [], 0
            if self.reducer is None:
                self.bucket = torch.zeros(total + 1, device=dev, dtype=torch.float32)
                for n, p in zip(names, params):
                    v = self.bucket[off:off + p.numel()].view(p.shape)
                    self.views[n] = v
                    self.view_list.append(v)
                    off += p.numel()
                self.t_sink = self.bucket[total:total + 1]
                self.t_view = self.t_sink.view(T.shape)
            else:
                # data parallel: the backward graph writes the reducer's own views -- its flat bucket is the collective's operand
                self.bucket = self.reducer.flat
                for n in names:
                    v = self.sink[n]
                    self.views[n] = v
                    self.view_list.append(v)
                self.t_view = next((v for p, v in self.reducer._autograd if p is T), None)
                if self.t_view is None:  # a frozen temperature: its (zero) gradient lands in a scratch word
                    self.t_view = torch.zeros(T.shape, device=dev, dtype=torch.float32)
                self.t_sink = self.t_view.view(-1)[:1]
            self.params = list(params)
            self.grad_pairs = list(zip(self.params, self.view_list))
            if T.requires_grad:
                self.grad_pairs.append((T, self.t_view))
            self._grad_params = [p for p, _ in self.grad_pairs]
            if self.reducer is None:
                for _, v in self.grad_pairs:  # (which plan a .grad tensor belongs to: commit_begin's mode 3)
                    _register_view(v, self)
            # the collective as the last node of the backward graph where the backend's launches can be captured (RCCL)
            self.collective_in_graph = bool(self.reducer is not None and self.reducer.collective and self.reducer._avg
                                            and net.__dict__.get("surface_collective", "graph") == "graph")
            self._capture(net, x)
        self.guard = _Guard(net)

    @property
    def reducer(self):
        return self._reducer_ref() if self._reducer_ref is not None else None

    def __del__(self):
        bury(self.__dict__.get("gf"), self.__dict__.get("gb"))

    # ---- the launches (eager warm-up, then captured) ------------------------------------------------------------------
    def _forward_body(self, net):
        B, N, M, K = self.shape
        x = self.x
        yo, saved = pointnet.forward_impl(net, x, True, skip_last=True, use_plan=False)
        fc4 = getattr(net, "fc%d" % net.num_fc_layers)  # the head's output layer (fc4 of the registration architecture)
        T = net.project._temperature
        # (yo: the classification sampler's BatchNorm on the head's output needs every cloud's row -- the head produced the queries
        #  itself, the scan reads them instead of computing the output layer in its waves)
        self.y = torch.empty(B, 3, M, device=self.dev, dtype=torch.float32) if yo is None else yo.view(B, 3, M)
        # (the reconstruction variant squares max(T, floor): the clamped value is a static tensor of the forward graph, its
        #  gradient gate a launch of the backward graph)
        floor = net.project._temperature_floor
        self.t_eff = None
        if floor is not None:
            self.t_eff = torch.clamp(T.detach(), min=floor)
            T = self.t_eff
        fc = (saved["zf"][-1], saved["cf"][-1], fc4.weight.detach(), fc4.bias.detach()) if yo is None else None
        sec = self.out_sec
        self.simp = self.outbuf[0:B * M * 3].view(B, M, 3)
        self.values = self.outbuf[2 * sec:2 * sec + 8]
        _, proj, state = ops.step_loss_forward(x, self.y, fc, T, K, self.min_sigma, 1.0, 0.0, self.weight, True, keys=self.keys,
                                               proj_out=self.outbuf[sec:sec + B * M * 3].view(B, M, 3))
        _, keys, qpart, qmax, G = state[5]
        self.dpsum = torch.empty(B, device=self.dev, dtype=torch.float32)
        check(lib.sn_surface_values_keys(B, N, M, G, ptr(keys), ptr(qpart), ptr(qmax), ptr(T.detach().reshape(1)), self.min_sigma,
                                         self.weight, ptr(self.y), ptr(self.simp), ptr(self.dpsum), ptr(self.values),
                                         stream_of(self.x)), "sn_surface_values_keys")
        self.saved, self.state, self.proj = saved, state, proj

    def _backward_body(self, net):
        B, N, M, K = self.shape
        T = net.project._temperature if self.t_eff is None else self.t_eff
        blob = None
        if pointnet.conv_stack_backward_supported(net, B, N):
            blob = ctypes.create_string_buffer(lib.sn_step_tail_bytes())
        cfg = (K, self.min_sigma, 1.0, 0.0, self.weight)
        res = ops.step_loss_backward(self.x, self.y, T, self.state, cfg, self.up_scalars[0:1], self.t_sink, blob, self.up_proj,
                                     grad_sigma=self.up_scalars[1:2])
        gQ = res[0]
        if self.with_simp:
            gQ = torch.add(gQ, self.up_simp.permute(0, 2, 1))  # (B,3,M): + the script's own gradient on the simplified cloud
        grads = pointnet.backward_impl(net, self.saved, gQ.view(B, -1), self.views, None, step_tail=blob)
        missing = [n for n in self.views if grads.get(n) is not self.views[n]]
        if missing:
            raise RuntimeError("surface: the backward did not write %s into the gradient bucket" % missing[:3])
        if self.t_eff is not None:  # d max(T, floor) / dT
            net.project._gate_floor_(self.t_sink)
        if self.collective_in_graph:
            self.reducer._all_reduce_mean(self.reducer.flat)  # captured: RCCL's kernel replays as the graph's last node
        self.bwd_keep = (res, grads)

    def _capture(self, net, x):
        cur = torch.cuda.current_stream(self.dev)
        bufs = [b for b in net.buffers()]
        if self.reducer is not None:
            # the plan's views ARE the reducer's live bucket: a plan built in the middle of an optimizer step (gradient accumulation
            # over micro-batches, a second plan for a contended configuration) must hand the bucket back as it found it -- the
            # warm-up backward (and, in-graph, its all-reduce) overwrites it
            bufs = bufs + [self.reducer.flat]
        kept = [b.clone() for b in bufs]
        side = torch.cuda.Stream(self.dev)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            # one eager pass of exactly these launches (every kernel of the graphs has run once; the step's persistent scratch
            # exists); the running statistics it advanced are put back -- the warm-up is not a training step
            self.x.copy_(x)
            self._forward_body(net)
            self._backward_body(net)
            for b, c in zip(bufs, kept):
                b.copy_(c)
        cur.wait_stream(side)
        torch.cuda.synchronize(self.dev)
        self.saved = self.state = self.bwd_keep = None
        _collect_before_capture()
        self.pool = torch.cuda.graph_pool_handle()
        self.gf, self.gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        # captured on the warm-up's stream (its per-stream scratch is found again: see engine._capture) -- unless a collective is in
        # play: the process group's watchdog polls events the warm-up recorded on that stream, which HIP refuses while it captures
        self.stream = side
        kw = {} if (self.reducer is not None and self.reducer.collective) else {"stream": side}
        with torch.cuda.graph(self.gf, pool=self.pool, capture_error_mode="thread_local", **kw):
            self._forward_body(net)
        with torch.cuda.graph(self.gb, pool=self.pool, capture_error_mode="thread_local", **kw):
            self._backward_body(net)
        self.v_lsimp, self.v_sigma = self.values[0], self.values[1]

    # ---- per step ---------------------------------------------------------------------------------------------------
    def busy(self):
        o = self.owner
        if o is None:
            return False
        if o() is None:  # the forward that owned the activations was dropped without a backward: its scan left the keys behind
            self.keys.zero_()
            self.owner = None
            return False
        return True

    def commit_begin_sink(self):
        """Data-parallel plan (the bucket is a FlatGradAllReducer's): -> old bucket to add after the replay, or None when this
        backward may overwrite the views (GradSink.direct(): first backward of a step, or every .grad dropped by zero_grad())."""
        if self.sink.direct():
            return None
        old = self.bucket.clone()
        # a slot whose parameter's .grad is NOT the bucket's view right now holds nothing of this step (the temperature after
        # optimizer.zero_grad(set_to_none=True): its slice still has the previous step's total, while this step's first
        # contribution sits in a tensor autograd made -- commit_end_sink folds that one in): it must not be added back
        base = self.bucket.data_ptr()
        for p, v in self.grad_pairs:
            g = p.grad
            if g is None or g.data_ptr() != v.data_ptr():
                off = (v.data_ptr() - base) // 4
                old[off:off + v.numel()].zero_()
        return old

    def commit_end_sink(self, old):
        if old is not None:
            self.bucket.add_(old)
        for p, v in self.grad_pairs:
            g = p.grad
            if g is not v:
                if g is not None and g.data_ptr() != v.data_ptr():
                    v.add_(g)  # somebody assigned another tensor as .grad: fold it in, take the slot back
                p.grad = v
        self.sink.written = True
        if self.collective_in_graph:
            self.reducer._graph_reduced = True  # reducer.reduce() after backward(): bookkeeping only

    def commit_begin(self):
        """-> (mode, old): 0 every .grad is None (the views become the gradients), 1 every .grad IS its view (accumulate into
        the bucket), 3 every .grad is the view of ONE other plan of this module (a second sampled cloud under the same loss: that
        plan's bucket takes the sum in one launch), 2 anything else (per parameter)."""
        n = len(self.grad_pairs)
        grads = [p.grad for p in self._grad_params]  # (one pass of attribute reads; the usual step -- every .grad dropped by
        #  zero_grad() -- is decided without looking at the views at all.  `is None`, never list.count(None): == on a tensor
        #  goes through torch's operator dispatch, microseconds apiece)
        if len([1 for g in grads if g is None]) == n:
            return 0, None
        nnone = nours = 0
        other = None
        nother = 0
        for g, (p, v) in zip(grads, self.grad_pairs):
            if g is None:
                nnone += 1
            elif g is v:
                nours += 1
            else:
                o = _VIEW_OWNER.get(id(g))
                op = o[1]() if o is not None and o[0]() is g else None
                if op is not None and (other is None or op is other):
                    other, nother = op, nother + 1
        if nother == n and len(other.grad_pairs) == n:
            return 3, other
        old = self.bucket.clone() if nours else None
        return (1 if nours == n else 2), old

    def commit_end(self, mode, old):
        if mode == 0:
            for p, v in self.grad_pairs:
                p.grad = v
        elif mode == 1:
            self.bucket.add_(old)
        elif mode == 3:
            old.bucket.add_(self.bucket)  # (.grad stay the other plan's views)
        else:
            off = 0
            for p, v in self.grad_pairs:
                g = p.grad
                if g is None:
                    p.grad = v
                elif g is v:
                    v.add_(old[off:off + v.numel()].view(v.shape))
                else:
                    p.grad = g + v
                off += v.numel()


class _SurfaceFunction(torch.autograd.Function):
    """simp (B,M,3), proj (B,M,3), L_simp (), sigma () = step(x); backward = the captured backward (see the module docstring)."""

    @staticmethod
    def forward(ctx, plan, net, x, _anchor):
        with _device_guard(plan.dev):
            plan.x.copy_(x, non_blocking=True)
            plan.gf.replay()
            out = plan.outbuf if plan.static_out else plan.outbuf.clone()  # (one launch: what the script keeps is its own)
        token = _Token()
        plan.owner = weakref.ref(token)
        ctx.plan, ctx.net, ctx.token = plan, net, token
        ctx.weight = None  # set by a getter that was asked for another loss weight than the captured one
        ctx.done = False
        ctx.set_materialize_grads(False)
        B, _, M, _ = plan.shape
        sec, n = plan.out_sec, B * M * 3
        if plan.static_out:  # (views of the plan's own block: handed out detached from it)
            return (out[0:n].view(B, M, 3).detach(), out[sec:sec + n].view(B, M, 3).detach(), out[2 * sec].detach(),
                    out[2 * sec + 1].detach())
        # (views of a tensor made inside this forward: nothing to detach from)
        return out[0:n].view(B, M, 3), out[sec:sec + n].view(B, M, 3), out[2 * sec], out[2 * sec + 1]

    @staticmethod
    def backward(ctx, g_simp, g_proj, g_lsimp, g_sigma):
        plan, net = ctx.plan, ctx.net
        if ctx.done:
            raise RuntimeError("samplenet_amd.surface: this forward was already backpropagated (its activations are static "
                               "buffers of a captured graph; call the sampler again instead of retaining the graph)")
        if plan.owner is None or plan.owner() is not ctx.token:
            raise RuntimeError("samplenet_amd.surface: the captured activations of this forward were overwritten")
        ctx.done = True
        B, N, M, K = plan.shape
        T = net.project._temperature
        with _device_guard(plan.dev):
            st = stream_of(plan.x)
            if g_proj is not None:
                g_proj = ops._f32c(g_proj)
            if g_lsimp is not None:
                g_lsimp = ops._f32c(g_lsimp)
            if g_sigma is not None:
                g_sigma = ops._f32c(g_sigma)
            if g_simp is not None and not plan.with_simp:
                # first sight of a gradient on the simplified cloud: this step runs the launches eagerly (below); the next
                # forward captures graphs that take it as an operand (no warm steps: the configuration is known)
                net.__dict__["_sn_surface_simp_grad"] = True
                for cfg in net.__dict__.get("_sn_surface", {}).values():
                    cfg.plans, cfg.seen, cfg.contended = [], WARM_STEPS, 0
            if (g_simp is None or plan.with_simp) and ctx.weight is None:
                if plan.with_simp:
                    if g_simp is not None:
                        plan.up_simp.copy_(ops._f32c(g_simp), non_blocking=True)
                        plan.up_simp_dirty = True
                    elif plan.up_simp_dirty:
                        plan.up_simp.zero_()
                        plan.up_simp_dirty = False
                check(lib.sn_surface_gather_upstream(B * M * 3, ptr(g_lsimp), ptr(g_sigma), ptr(g_proj), ptr(plan.up_scalars),
                                                     ptr(plan.up_proj), st), "sn_surface_gather_upstream")
                if plan.reducer is not None:
                    old = plan.commit_begin_sink()
                    plan.gb.replay()
                    plan.commit_end_sink(old)
                else:
                    mode, old = plan.commit_begin()
                    plan.gb.replay()
                    plan.commit_end(mode, old)
            else:
                # irregular upstream (a gradient on the simplified cloud itself / another loss weight): the same launches,
                # eagerly, on the graph's activations
                zero = torch.zeros(1, device=plan.dev)
                gl = g_lsimp.reshape(1) if g_lsimp is not None else zero
                gs = g_sigma.reshape(1) if g_sigma is not None else zero
                gp = g_proj if g_proj is not None else torch.zeros(B, M, 3, device=plan.dev)
                w = plan.weight if ctx.weight is None else ctx.weight
                res = ops.step_loss_backward(plan.x, plan.y, T if plan.t_eff is None else plan.t_eff, plan.state,
                                             (K, plan.min_sigma, 1.0, 0.0, w), gl, None, None, gp, grad_sigma=gs)
                gQ, gT = res[0], res[1]
                if plan.t_eff is not None:
                    gT = gT * (T.detach().reshape(gT.shape) >= net.project._temperature_floor).to(gT.dtype)
                if g_simp is not None:
                    gQ = gQ + ops._f32c(g_simp).permute(0, 2, 1)
                grads = pointnet.backward_impl(net, plan.saved, gQ.reshape(B, -1).contiguous(), None, None)
                if plan.reducer is not None:
                    plan.sink.commit(grads)  # (torch semantics on the reducer's views; no collective: reducer.reduce() follows,
                    #  GradSink.commit marks the bucket as holding an unreduced contribution)
                    if T.requires_grad:
                        plan.t_view.add_(gT.reshape(T.shape)) if T.grad is not None else plan.t_view.copy_(gT.reshape(T.shape))
                        T.grad = plan.t_view
                else:
                    for n, p in zip(pointnet.param_order(net), plan.params):
                        p.grad = grads[n] if p.grad is None else p.grad + grads[n]
                    if T.requires_grad:
                        gT = gT.reshape(T.shape)
                        T.grad = gT.clone() if T.grad is None else T.grad + gT
        plan.owner = None
        return None, None, None, None


def _supported(net, x):
    if not (getattr(net, "graph_surface", True) and net.use_hip_mlp) or net.skip_projection:
        return False
    if not getattr(net, "standard_arch", False) and not _variant_ok(net):
        return False
    if net.input_shape != "bnc" or net.output_shape != "bnc":
        return False
    if not torch.is_grad_enabled() or not x.is_cuda or x.requires_grad or x.dtype != torch.float32 or x.dim() != 3 or x.shape[2] != 3:
        return False
    d = net.__dict__
    if d.get("_sn_sync_bn") is not None:
        return False
    sink = d.get("_grad_sink")
    if sink is not None:
        red = getattr(sink, "reducer", None)  # a FlatGradAllReducer's sink: the plan writes the reducer's bucket
        if red is None:
            return False
        after = d.get("_after_fc_grads")
        if after is not None and getattr(after, "__self__", None) is not red:
            return False
    elif d.get("_after_fc_grads") is not None:
        return False
    if net._forward_hooks or net._forward_pre_hooks or net._backward_hooks:
        return False
    why = autograd_listeners(net)
    if why is not None:
        return _fallback(net, why)
    return True


def _variant_ok(net):
    """The sampler variants of the TF packages besides the registration architecture: any conv / FC widths and FC layers without
    BatchNorm (reconstruction/src/samplers.py:23-38) run the same launches -- the MLP entry points are generic over the layer
    list, the scan computes its queries from the last hidden layer whatever its width.  The classification sampler's BatchNorm
    on the head's OUTPUT (samplenet_model.py:30-108) needs all clouds' queries before any scan can start: the head then ends in
    sn_layer_forward_bn_out and the scan reads the queries (pointnet.out_bn; one launch more in either graph).  A last BatchNorm
    that torch applies (SyncBatchNorm) stays op by op."""
    ok = net.__dict__.get("_sn_variant_ok")
    if ok is None:
        _, fcs = pointnet._layers(net)
        last_bn = net._modules.get("bn_fc%d" % net.num_fc_layers)
        ob = pointnet.out_bn(net)
        hip_last = last_bn is None or (ob is not None and ob[1] is last_bn and last_bn.momentum is not None)
        ok = net.__dict__["_sn_variant_ok"] = bool(len(fcs) >= 2 and hip_last and fcs[-1].bn is None and fcs[-2].Co % 4 == 0
                                                   and "_features" not in net.__dict__)
    return ok


def autograd_listeners(net):
    """Why the captured surface must step aside for this module, or None.  Captured gradients land in .grad without
    AccumulateGrad nodes running, so anything that listens on those needs the op-by-op route: per-parameter hooks, and torch
    DistributedDataParallel (its reducer hangs on the AccumulateGrad nodes, invisible from a child module -- assumed whenever
    a multi-rank process group exists and the module has no FlatGradAllReducer of its own)."""
    if _has_param_hooks(net):
        return "a parameter carries an autograd hook"
    if (net.__dict__.get("_grad_sink") is None and getattr(net, "graph_surface", True) != "force" and _dist.is_available()
            and _dist.is_initialized() and _dist.get_world_size() > 1):
        return ("torch.distributed runs %d ranks and no FlatGradAllReducer is attached (DistributedDataParallel's reducer hooks "
                "would never fire on captured gradients; graph_surface='force' overrides)" % _dist.get_world_size())
    return None


def _has_param_hooks(net):
    ps = net.__dict__.get("_sn_hook_params")
    sig = tuple(map(id, net._modules.values()))  # (a REPLACED child keeps the count: key on the child objects themselves)
    if ps is None or ps[0] != sig:
        ps = net.__dict__["_sn_hook_params"] = (sig, [p for p in net.parameters()])
    return any(map(_bw_hooks, ps[1])) or any(map(_pa_hooks, ps[1]))


_bw_hooks = operator.attrgetter("_backward_hooks")
_pa_hooks = (operator.attrgetter("_post_accumulate_grad_hooks") if hasattr(torch.Tensor, "_post_accumulate_grad_hooks")
             else (lambda p: None))


def _fallback(net, why):
    if not net.__dict__.get("_sn_surface_warned"):
        net.__dict__["_sn_surface_warned"] = True
        warnings.warn("samplenet_amd.surface: %s -- this module runs op by op (no captured graphs)" % why)
    return False


class _Config:
    """Per (batch shape, device): the warm-step count, the captured plans, how often a forward found all of them taken."""

    __slots__ = ("seen", "plans", "contended", "dead", "used")

    def __init__(self):
        self.seen, self.plans, self.contended, self.dead, self.used = 0, [], 0, False, 0


_clock = 0


def _make_room(table, keep):
    """Before a configuration captures its first graphs: at most MAX_CONFIGS - 1 others keep theirs (every plan holds the step's
    activations, ~50 MB at 32 x 1024 points) -- the least recently used one whose plans are all idle starts over (a script that
    walks through many batch shapes would otherwise pin a set of buffers per shape)."""
    holders = [c for c in table.values() if c is not keep and c.plans]
    while len(holders) >= MAX_CONFIGS:
        idle = [c for c in holders if not any(p.busy() for p in c.plans)]
        if not idle:
            return
        victim = min(idle, key=lambda c: c.used)
        victim.plans, victim.seen, victim.contended = [], 0, 0
        holders.remove(victim)


def _build(net, x):
    from .fused_step import external_task_supported

    net.__dict__.pop("_sn_hook_params", None)  # (the parameter objects the hook check walks: re-read at every capture)
    if autograd_listeners(net) is not None:
        return None

    T = net.project._temperature
    params = pointnet.param_list(net)
    ok = (external_task_supported(net, x) and all(p.requires_grad for p in params) and T.dim() == 0 and
          all(L.bn is None or (L.bn.momentum is not None and L.bn.track_running_stats)
              for L in sum(pointnet._layers(net), [])[:-1]))
    sink = net.__dict__.get("_grad_sink")
    if ok and sink is not None:  # every gradient the backward graph writes must be a view of the reducer's bucket
        red = sink.reducer
        ok = all(n in sink for n in pointnet.param_order(net)) and (not T.requires_grad or any(p is T for p, _ in red._autograd))
    if not ok:
        return None
    red = getattr(sink, "reducer", None) if sink is not None else None
    in_graph = bool(red is not None and red.collective and red._avg and net.__dict__.get("surface_collective", "graph") == "graph")
    plan, err = None, None
    try:
        plan = _Plan(net, x, net.__dict__.get("_sn_surface_weight", 1.0))
    except Exception as e:  # noqa: BLE001 -- a configuration that cannot be captured stays on the op-by-op route
        err = e
        torch.cuda.synchronize(x.device)
    if in_graph:
        # the collective inside the backward graph: every rank must end up in the SAME mode -- a rank whose capture failed and
        # fell back to reducer.reduce() launching the all-reduce, beside ranks whose graphs carry it, would issue a different
        # number of collectives per step.  One flag, minimum over the ranks (plans are built at the same step on
        # every rank: the warm-step count is per configuration and the ranks run the same script).
        flag = torch.tensor([1 if plan is not None else 0], device=x.device, dtype=torch.int32)
        if red.world > 1:
            _dist.all_reduce(flag, op=_dist.ReduceOp.MIN, group=red.group)
        if int(flag.item()) == 0:
            warnings.warn("samplenet_amd.surface: capturing the gradient all-reduce inside the backward graph failed on %s (%s); "
                          "reducer.reduce() will launch it after backward() on every rank"
                          % ("this rank" if plan is None else "another rank", repr(err)[:200]))
            net.__dict__["surface_collective"] = "after"
            plan, err = None, None
            try:
                plan = _Plan(net, x, net.__dict__.get("_sn_surface_weight", 1.0))
            except Exception as e2:  # noqa: BLE001
                err = e2
                torch.cuda.synchronize(x.device)
    if plan is None:
        warnings.warn("samplenet_amd.surface: capture failed, this configuration stays op by op (%s)" % repr(err)[:300])
    return plan


def plans(net):
    """The captured plans of a module (all configurations)."""
    return [p for cfg in net.__dict__.get("_sn_surface", {}).values() for p in cfg.plans]


def _lives(net):
    """Live records of the forwards that still wait for their backward (pruned of finished / dropped ones)."""
    lives = net.__dict__.get("_sn_surface_live")
    if not lives:
        return []
    lives = [lv for lv in lives if lv.plan.owner is not None and lv.token() is not None and lv.plan.owner() is lv.token()]
    net.__dict__["_sn_surface_live"] = lives
    return lives


def _demote_lives(net):
    for lv in net.__dict__.get("_sn_surface_live") or ():
        lv.demote()


def try_forward(net, x):
    """-> (simp, proj) from the captured forward, or None: the caller runs the op-by-op forward."""
    if not ENABLED or _suspend or not _supported(net, x):
        return None
    if torch.cuda.is_current_stream_capturing():  # somebody captures the step themselves: plain launches for their graph
        return None
    if _GRAVE:
        flush_grave()
    _demote_lives(net)  # (earlier forwards keep weak records only: dropped outputs free their plan)
    table = net.__dict__.setdefault("_sn_surface", {})
    key = (x.shape[0], x.shape[1], x.device)
    cfg = table.get(key)
    if cfg is None:
        cfg = table[key] = _Config()
    if cfg.dead:
        return None
    global _clock
    _clock += 1
    cfg.used = _clock
    if cfg.plans and not cfg.plans[0].guard.ok():
        # a parameter / buffer / layer was replaced: new graphs after the warm steps (this call is the first of them)
        cfg.plans, cfg.seen, cfg.contended = [], 1, 0
        net.__dict__.pop("_sn_hook_params", None)
        return None
    if not cfg.plans:
        cfg.seen += 1
        if cfg.seen <= WARM_STEPS:
            return None
        _make_room(table, cfg)
        plan = _build(net, x)
        if plan is None:
            cfg.dead = True
            return None
        cfg.plans.append(plan)
    plan = next((p for p in cfg.plans if not p.busy()), None)
    if plan is None:
        # every plan's activations wait for a backward: a script that samples several clouds under one loss (registration/
        # main.py:516-524 -- its default).  Seen WARM_STEPS times, the configuration gets one more pair of graphs (up to MAX_PLANS)
        cfg.contended += 1
        if cfg.contended <= WARM_STEPS or len(cfg.plans) >= MAX_PLANS:
            return None
        plan = _build(net, x)
        if plan is None:
            cfg.contended = -(1 << 30)
            return None
        cfg.plans.append(plan)
        cfg.contended = 0
    # (one differentiable input is enough to make the outputs differentiable: the gradients do not travel through autograd)
    simp, proj, lsimp, sigma = _SurfaceFunction.apply(plan, net, x, plan.params[0])
    live = _Live()
    live.weak = False
    live.x, live.x_version, live._simp, live._lsimp, live._sigma, live.plan = x, x._version, simp, lsimp, sigma, plan
    live.t_version = net.project._temperature._version
    live.token = plan.owner
    net.__dict__["_sn_surface_live"] = _lives(net) + [live]
    return simp, proj


class _ReweightFunction(torch.autograd.Function):
    """L_simp for another weight than the captured one, from the captured components; tells the node's backward."""

    @staticmethod
    def forward(ctx, lsimp, value, node_ctx, weight):
        ctx.node_ctx, ctx.weight = node_ctx, weight
        return value.clone()

    @staticmethod
    def backward(ctx, g):
        ctx.node_ctx.weight = ctx.weight
        return g, None, None, None


def simplification_loss(net, ref_pc, samp_pc, weight):
    """The captured forward's L_simp when (ref_pc, samp_pc) are the tensors that forward consumed / returned, else None."""
    live = next((lv for lv in _lives(net) if samp_pc is lv.simp), None)
    if live is None:
        return None
    x, lsimp = live.cloud, live.lsimp
    if x is None or lsimp is None:
        return None
    same = ref_pc is x or (ref_pc.data_ptr() == x.data_ptr() and ref_pc.shape == x.shape and ref_pc.stride() == x.stride())
    if not same or x._version != live.x_version:
        return None
    weight = float(weight)
    net.__dict__["_sn_surface_weight"] = weight
    if weight == live.plan.weight:
        return lsimp
    # another loss weight than the captured one: value from the components, eager backward this step, new graphs afterwards
    v = live.plan.values
    value = v[2] + v[3] + weight * v[4]
    for cfg in net.__dict__.get("_sn_surface", {}).values():
        if live.plan in cfg.plans:
            cfg.plans, cfg.seen, cfg.contended = [], WARM_STEPS, 0  # recaptured (with this weight) by the next forward
    return _ReweightFunction.apply(lsimp, value, lsimp.grad_fn, weight)  # (a Function's ctx IS its outputs' grad_fn)


def projection_loss(net):
    """sigma as an output of the most recent captured forward that still waits for its backward (any of them carries the same
    value; the gradient reaches the temperature through that node)."""
    lives = _lives(net)
    if not lives or net.project._temperature._version != lives[-1].t_version:
        return None
    return lives[-1].sigma  # (None when only a weak record is left: the caller takes sigma() through autograd)
