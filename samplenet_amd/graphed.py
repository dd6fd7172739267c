"""Captured work behind a FROZEN network's call: the task side of the reference's training step.

registration/main.py:557-577 evaluates the frozen task network on the sampled cloud in every sampler step
(twist = model(p0, p1); p1_est = rotate(p0); Chamfer(p1, p1_est)): ~32 launches behind ~0.5 ms of Python and autograd for
0.27 ms of GPU work.  `call(owner, tag, fn, args)` runs such a pure function of tensors -- no parameter of `owner` wants a
gradient, the outputs depend on nothing but the arguments and the (frozen) parameters -- on two hipGraphs per configuration
once it has been seen WARM_STEPS times, the way torch.cuda.make_graphed_callables would, but transparently:

    forward graph   fn(*static copies of the arguments)                      -> static outputs
    backward graph  torch.autograd.grad(outputs, arguments, static upstream) -> static argument gradients

behind one autograd node.  Anything irregular runs fn eagerly: a second call before the first one's backward, another
thread's capture in progress, a parameter that was replaced or unfrozen (guard), no_grad.  Parameters updated IN PLACE are
seen by the replays (the kernels read them where they lie; bf16 weight planes are re-split inside a captured call,
task_features._weight_planes).  Outputs and argument gradients are STATIC tensors, overwritten by the next call of the same
configuration (as with make_graphed_callables): clone what must outlive a step.
"""
import weakref

import torch

from . import surface

WARM_STEPS = 2
MAX_PLANS = 6  # captured configurations per owner: the least recently used idle one gives its graphs and static buffers up
_clock = 0


class _Token:
    __slots__ = ("__weakref__",)


def _owned_modules(owner):
    """The modules whose parameters a call of `owner` reads: its tree WITHOUT the children it merely carries for the script --
    registration/main.py:296 attaches the (trainable) sampler to the task network as `model.sampler`, which PCRNet.forward
    never calls.  owner._graphed_exclude names such children: OPT-IN per owner class (task_features.PCRNet sets ("sampler",)); any
    other owner's children are all owned -- a trainable child that the graphed call does execute is then seen by the
    frozen-parameter check and by the guard."""
    skip = set(getattr(owner, "_graphed_exclude", ()))
    out, stack = [], [(owner, True)]
    while stack:
        m, top = stack.pop()
        out.append(m)
        for name, child in m._modules.items():
            if child is not None and not (top and name in skip):
                stack.append((child, False))
    return out


def _owned_parameters(owner):
    for m in _owned_modules(owner):
        for p in m._parameters.values():
            if p is not None:
                yield p


class _ModuleGuard:
    """Every parameter / buffer of the modules the call reads (_owned_modules) is still the object at the address the graphs
    read, and still frozen; the children are still the same objects."""

    def __init__(self, module):
        self.mods, self.tens = [], []
        skip = set(getattr(module, "_graphed_exclude", ()))
        for m in _owned_modules(module):
            for name, child in m._modules.items():
                if not (m is module and name in skip):
                    self.mods.append((m._modules, name, child))
            for d in (m._parameters, m._buffers):
                for k, t in d.items():
                    if t is not None:
                        self.tens.append((d, k, t, t.data_ptr(), t._version))

    def ok(self):
        for d, k, m in self.mods:
            if d.get(k) is not m:
                return False
        # (the version counter: the graphs were captured with the frozen weights' bf16 planes split OUTSIDE them -- task_features.
        #  static_capture -- so a weight written in place since (load_state_dict, an optimizer step) needs new graphs)
        for d, k, t, p, v in self.tens:
            if d.get(k) is not t or t.data_ptr() != p or t.requires_grad or t._version != v:
                return False
        return True


def _flat(out):
    return tuple(out) if isinstance(out, (tuple, list)) else (out,)


class _Plan:
    def __init__(self, owner, fn, args):
        dev = args[0].device
        self.dev = dev
        self.owner_token = None
        with torch.cuda.device(dev):
            self.ins = [torch.empty(a.shape, device=dev, dtype=a.dtype).requires_grad_(a.requires_grad) for a in args]
            self.ins_req = [s for s in self.ins if s.requires_grad]
            cur = torch.cuda.current_stream(dev)
            side = torch.cuda.Stream(dev)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                with torch.no_grad():
                    for s, a in zip(self.ins, args):
                        s.copy_(a)
                for _ in range(2):  # eager passes of exactly these launches: lazy caches (constant tables, weight planes) exist
                    outs = _flat(fn(*self.ins))
                    req = [o for o in outs if o.requires_grad]
                    torch.autograd.grad(req, self.ins_req, [torch.ones_like(o) for o in req], allow_unused=True)
                del outs, req
            cur.wait_stream(side)
            torch.cuda.synchronize(dev)
            surface._collect_before_capture()
            self.pool = torch.cuda.graph_pool_handle()
            self.gf, self.gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            self.guard = _ModuleGuard(owner)  # (before the capture: the versions the graphs' constant weight planes belong to)
            # captured on the stream the eager passes ran on: the per-stream scratch they created (skinny counters, constant
            # tables) is found again -- no allocation + fill launches inside the graphs; the owner's frozen weights are constants
            # of these graphs (their bf16 planes exist since the eager passes: no split launches either)
            from .task_features import static_capture

            self.stream = side
            with static_capture(list(_owned_parameters(owner))):
                with torch.cuda.graph(self.gf, pool=self.pool, stream=side, capture_error_mode="thread_local"):
                    out = fn(*self.ins)
                self.single = not isinstance(out, (tuple, list))
                self.outs = _flat(out)
                self.req = [i for i, o in enumerate(self.outs) if o.requires_grad]
                self.gouts = [torch.zeros_like(self.outs[i]) for i in self.req]
                with torch.cuda.graph(self.gb, pool=self.pool, stream=side, capture_error_mode="thread_local"):
                    self.gins = torch.autograd.grad([self.outs[i] for i in self.req], self.ins_req, self.gouts, allow_unused=True)
        self.used = 0

    def __del__(self):
        surface.bury(self.__dict__.get("gf"), self.__dict__.get("gb"))  # (never destroyed during somebody's capture: surface.py)

    def busy(self):
        o = self.owner_token
        if o is None:
            return False
        if o() is None:
            self.owner_token = None
            return False
        return True


class _GraphedFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, plan, *args):
        with torch.cuda.device(plan.dev), torch.no_grad():
            for s, a in zip(plan.ins, args):
                if a.data_ptr() != s.data_ptr():
                    s.copy_(a, non_blocking=True)
            plan.gf.replay()
        token = _Token()
        plan.owner_token = weakref.ref(token)
        ctx.plan, ctx.token, ctx.done = plan, token, False
        outs = tuple(o.detach() for o in plan.outs)
        ctx.mark_non_differentiable(*[o for i, o in enumerate(outs) if i not in plan.req])
        ctx.set_materialize_grads(False)
        return outs

    @staticmethod
    def backward(ctx, *gouts):
        plan = ctx.plan
        if ctx.done or plan.owner_token is None or plan.owner_token() is not ctx.token:
            raise RuntimeError("samplenet_amd.graphed: this call was already backpropagated or its captured activations were "
                               "overwritten (they are static buffers of a graph; evaluate the network again)")
        ctx.done = True
        with torch.cuda.device(plan.dev), torch.no_grad():
            for i, sg in zip(plan.req, plan.gouts):
                g = gouts[i]
                if g is None:
                    sg.zero_()
                else:
                    sg.copy_(g, non_blocking=True)
            plan.gb.replay()
        plan.owner_token = None
        it = iter(plan.gins)
        return (None,) + tuple((next(it) if s.requires_grad else None) for s in plan.ins)


def call(owner, tag, fn, args):
    """fn(*args) on the captured graphs of this configuration, or None: the caller runs fn(*args) itself.
    owner: the frozen nn.Module whose parameters fn reads (the plans live in its __dict__); args: CUDA tensors, at least one of
    which wants a gradient; fn must be a pure function of them."""
    if not surface.ENABLED or surface._suspend or not getattr(owner, "graph_surface", True) or not torch.is_grad_enabled():
        return None
    dev = None
    any_grad = False
    for a in args:
        # (views are fine -- prefixes of a projected cloud: the arguments are copied into the graphs' static inputs anyway)
        if not isinstance(a, torch.Tensor) or not a.is_cuda or (dev is not None and a.device != dev):
            return None
        dev = a.device
        any_grad = any_grad or a.requires_grad
    if not any_grad or torch.cuda.is_current_stream_capturing():
        return None
    if surface._GRAVE:
        surface.flush_grave()
    table = owner.__dict__.setdefault("_sn_graphed", {})
    key = (tag,) + tuple((tuple(a.shape), a.dtype, a.device, a.requires_grad) for a in args)
    ent = table.get(key)
    if ent is False:
        return None
    if not isinstance(ent, _Plan):
        n = (ent or 0) + 1
        if n <= WARM_STEPS:
            table[key] = n
            return None
        if any(p.requires_grad for p in _owned_parameters(owner)) or owner._forward_hooks or owner._forward_pre_hooks:
            table[key] = 0
            return None
        held = [(k, v) for k, v in table.items() if isinstance(v, _Plan)]
        if len(held) >= MAX_PLANS:  # (a caller that walks through many shapes must not pin a set of buffers per shape)
            idle = [(k, v) for k, v in held if not v.busy()]
            if idle:
                table[min(idle, key=lambda kv: kv[1].used)[0]] = 0
        try:
            with surface.suspended():  # (nothing inside the captured call builds graphs of its own)
                ent = _Plan(owner, fn, args)
        except Exception as e:  # noqa: BLE001 -- a configuration that cannot be captured stays op by op
            import warnings

            warnings.warn("samplenet_amd.graphed: capture of %r failed, it stays op by op (%s)" % (tag, repr(e)[:300]))
            torch.cuda.synchronize(dev)
            ent = False
        table[key] = ent
        if ent is False:
            return None
    plan = ent
    global _clock
    _clock += 1
    plan.used = _clock
    if plan.busy():
        return None
    if not plan.guard.ok():
        table[key] = 1
        return None
    outs = _GraphedFunction.apply(plan, *args)
    return outs[0] if plan.single else outs
