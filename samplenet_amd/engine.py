"""Whole-step execution of the sampler's training step as ONE hipGraph.

At the reference's batch size (B = 32) the step is a chain of ~70 short kernels; launched one by one from
Python it is host-bound (a few microseconds of launch path per kernel, tens of microseconds of Python per op).
`SamplerTrainStep` captures   forward -> losses -> backward   once (torch.cuda.CUDAGraph: the ctypes-launched HIP
kernels are recorded like any other work on the capturing stream) and replays it per batch; inputs are copied
into a static buffer, gradients land in static tensors (or in the flat all-reduce bucket).

The loss is the one registration/main.py:507-531 builds for the sampler:
    alpha * simplification_loss + lmbda * projection_loss + task_loss(proj)
with `task_loss` a callable (default: mean of the projected points -- the stand-in SURVEY.md 8d prescribes
for the benchmark so that the projection branch receives gradient).
"""
import torch


class SamplerTrainStep:
    def __init__(self, net, example_x, alpha=0.01, lmbda=0.01, gamma=1.0, delta=0.0, task_loss=None, reducer=None,
                 use_graph=True, warmup=3, fused_loss=True, input_ring=None, fused_head=True):
        self.net, self.reducer = net, reducer
        self.alpha, self.lmbda, self.gamma, self.delta = alpha, lmbda, gamma, delta
        self.fused_loss = fused_loss  # False: compose the loss op by op through the module's own methods (A/B, tests)
        self.fused_head = fused_head  # fast path only: fc4's forward computed inside the pair scan (fused_step.py)
        self.task_loss = task_loss  # None: the benchmark's stand-in mean(proj), fused with the loss weighting
        # input_ring: optional list of device tensors (shape of example_x) that the caller fills IN PLACE -- e.g. the
        # host-to-device targets of its data loader.  One graph is captured per entry (all sharing one memory pool) and
        # replay(i) runs the step on entry i without the copy into a static buffer that __call__(x) needs.
        self.ring = list(input_ring) if input_ring is not None else None
        if self.ring is not None and use_graph and reducer is None:
            raise ValueError("input_ring needs a gradient sink (FlatGradAllReducer): the graphs must write one set of .grad tensors")
        self.x = example_x.clone() if self.ring is None else self.ring[0]
        self._one = torch.ones((), device=example_x.device, dtype=torch.float32)
        self.graph = None
        self.loss = None
        self._ring_graphs, self._ring_loss = [], []
        if use_graph and reducer is not None:
            reducer.disable_overlap()  # the graph replays backward without Python: one all-reduce after the replay
        if use_graph:
            self._capture(warmup)

    def _fast_path(self):
        """forward + loss behind one autograd node (ops.SamplerStepLossFunction): the benchmark's stand-in task term,
        training mode with projection, (B,N,3) input, and a batch small enough that the pair scan splits clouds."""
        net = self.net
        if not self.fused_loss or self.task_loss is not None or not net.training or net.skip_projection or \
                net.input_shape != "bnc":
            return False
        from ._lib import lib

        B, N, _ = self.x.shape
        return lib.sn_pairscan_colmin_splits(B, N, net.num_out_points) > 1

    def _step(self):
        net, x = self.net, self.x
        if self._fast_path():
            from . import ops

            T = net.project._temperature
            t_sink = None
            if self.reducer is not None and T.requires_grad and T.grad is not None:
                t_sink = T.grad.view(-1)[:1]  # a view of the flat bucket: written in place, nothing to zero or accumulate
            elif self.reducer is not None:
                self.reducer.zero_grad()
            weight = self.gamma + self.delta * net.num_out_points
            if self.fused_head and net.use_hip_mlp:
                from .fused_step import sampler_step

                loss, _y, _proj = sampler_step(net, x, self.alpha, self.lmbda, weight, t_sink, True)  # fc4 inside the scan
            else:
                y = net._features(x.permute(0, 2, 1), x)  # (B,3,M)
                loss, _proj = ops.SamplerStepLossFunction.apply(y, x, T, net.project._group_size, net.project._min_sigma_f,
                                                                self.alpha, self.lmbda, weight, t_sink, True)
            loss.backward(self._one)  # preallocated upstream gradient: no ones_like fill per step
            return loss.detach()
        if self.reducer is not None:
            self.reducer.zero_grad()
        simp, proj = net(x)
        lsimp = net.get_simplification_loss(x, simp, net.num_out_points, self.gamma, self.delta)
        if self.task_loss is None and net.training and not net.skip_projection:
            # alpha * L_simp + lmbda * sigma + mean(proj) in one fused kernel pair (same value as the composition below)
            from . import ops

            loss = ops.SamplerLossFunction.apply(lsimp, net.project._temperature, proj, self.alpha, self.lmbda,
                                                 net.project._min_sigma_f)
        else:
            task = self.task_loss(proj) if self.task_loss is not None else proj.mean()
            loss = self.alpha * lsimp + self.lmbda * net.get_projection_loss() + task
        loss.backward(self._one)  # preallocated upstream gradient: no ones_like fill per step
        return loss.detach()

    def _capture(self, warmup):
        if self.reducer is None:
            for p in self.net.parameters():
                p.grad = None
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self._step()
                if self.reducer is None:
                    for p in self.net.parameters():
                        p.grad = None
        torch.cuda.current_stream().wait_stream(side)
        if self.ring is None:
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
                self.loss = self._step()
            return
        pool = torch.cuda.graph_pool_handle()
        for buf in self.ring:
            if buf.shape != self.ring[0].shape or buf.device != self.ring[0].device or not buf.is_contiguous():
                raise ValueError("input_ring entries must be contiguous tensors of one shape on one device")
            self.x = buf
            g = torch.cuda.CUDAGraph()
            # thread_local: API calls of other threads (the RCCL watchdog polling events) must not invalidate the capture
            with torch.cuda.graph(g, pool=pool, capture_error_mode="thread_local"):
                loss = self._step()
            self._ring_graphs.append(g)
            self._ring_loss.append(loss)
        self.graph = self._ring_graphs[0]
        self.loss = self._ring_loss[0]

    def replay(self, i):
        """One step on ring entry i (whatever the caller wrote into input_ring[i]); returns the (static) loss tensor."""
        if self._ring_graphs:
            self._ring_graphs[i].replay()
            self.loss = self._ring_loss[i]
        else:
            self.x = self.ring[i]
            self.loss = self._step()
        if self.reducer is not None:
            self.reducer.reduce()
        return self.loss

    def __call__(self, x):
        """Runs one step on batch x (same shape as example_x); returns the (static) loss tensor.
        Gradients are in p.grad afterwards (cross-rank averaged when a reducer is attached)."""
        if self.ring is not None:
            raise RuntimeError("this step was built on an input ring: fill input_ring[i] in place and call replay(i)")
        self.x.copy_(x, non_blocking=True)
        if self.graph is not None:
            self.graph.replay()
        else:
            self.loss = self._step()
        if self.reducer is not None:
            self.reducer.reduce()
        return self.loss
