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
                 use_graph=True, warmup=3, fused_loss=True, input_ring=None, fused_head=True, overlap_allreduce=None,
                 allreduce="graph"):
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
        if net.__dict__.get("_sn_sync_bn") is not None and use_graph:
            # (statistics over all ranks: host-ordered collectives between the layers -- the step runs eagerly)
            if self.ring is not None:
                raise ValueError("input_ring needs the captured step; synchronised BatchNorm runs eagerly")
            use_graph = False
        self.x = example_x.clone() if self.ring is None else self.ring[0]
        self._one = torch.ones((), device=example_x.device, dtype=torch.float32)
        self.graph = None
        self.loss = None
        self._ring_graphs, self._ring_loss, self._ring_outputs = [], [], []
        self.outputs = None  # (simplified, projected) clouds of the last step (static tensors under graph replay)
        if use_graph and reducer is not None:
            reducer.disable_overlap()  # the graph replays backward without Python: the engine places the collectives
        # Captured step + gradient collective (N > 1): the step is captured as TWO graphs split where the FC head's gradients
        # (86 % of the bucket) are final; between the replays their all-reduce starts on the reducer's side stream and runs
        # beside the conv stack's backward (graph 2); the rest of the bucket follows graph 2.  Stream-ordered: no host sync.
        if overlap_allreduce is None:
            overlap_allreduce = False
        self.split = bool(use_graph and overlap_allreduce and reducer is not None and reducer.collective and self._fast_path()
                          and task_loss is None and fused_head and getattr(net, "use_hip_mlp", False)
                          and getattr(net, "standard_arch", True))
        # Where the gradient collective of a captured step runs (N > 1):
        #   "graph"      (default) INSIDE the step's one graph, at its end on the capturing stream: RCCL's kernel is one more
        #                node of the replay -- no host-side launch path per step, nothing between the graph and the collective;
        #   "graph-fork" inside the graph, the FC-head segment (86 % of the bytes) forked onto the reducer's side stream where
        #                those gradients are final and joined at the end: hidden under the conv stack's backward at the price of
        #                two cross-stream edges in the graph;
        #   "after"      one collective launched from Python behind each replay (round-2 behaviour).
        #   "auto"       the three placements above are each captured on the first ring entry and timed (20 replays, the
        #                slowest rank's time counts), the fastest one is kept: which of them wins depends on what the collective
        #                costs beside the step's own kernels at THIS world size (a ~1 MB ring all-reduce over xGMI is latency-
        #                bound: tens of microseconds against a 0.18 ms step) -- unknowable from a single-GPU run.  The choice and
        #                the three times are left in self.allreduce_probe.
        if allreduce not in ("graph", "graph-fork", "after", "auto"):
            raise ValueError("allreduce: 'graph', 'graph-fork', 'after' or 'auto'")
        self.allreduce_probe = None
        auto = allreduce == "auto" and bool(use_graph and not self.split and reducer is not None and reducer.collective)
        if allreduce == "auto":
            allreduce = "graph"
        self.allreduce = allreduce
        self.in_graph = bool(use_graph and not self.split and reducer is not None and reducer.collective and allreduce != "after")
        if use_graph and auto:
            self._choose_allreduce(warmup)
        elif use_graph:
            self._capture(warmup)

    def _choose_allreduce(self, warmup, replays=20):
        """allreduce='auto': see __init__."""
        import torch.distributed as dist

        from .surface import bury

        ring = self.ring
        times = {}
        for mode in ("graph", "after", "graph-fork"):
            self.allreduce, self.in_graph = mode, mode != "after"
            self._ring_graphs, self._ring_loss, self._ring_outputs = [], [], []
            if ring is not None:
                self.ring = ring[:1]
            ms = float("inf")
            multi = dist.is_initialized() and dist.get_world_size(self.reducer.group) > 1
            try:
                ok = True
                try:
                    self._capture(warmup)
                except RuntimeError:
                    torch.cuda.synchronize()
                    ok = False
                ok = ok and self.allreduce == mode  # (a capture that fell back to 'after' is not a measurement of `mode`)
                if multi:  # the timed replays below issue collectives: every rank runs them, or none does
                    flag = torch.tensor([1 if ok else 0], device=self.x.device, dtype=torch.int32)
                    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.reducer.group)
                    ok = bool(int(flag.item()))
                if ok:
                    for _ in range(3):
                        self._probe_step()
                    torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(replays):
                        self._probe_step()
                    e1.record()
                    torch.cuda.synchronize()
                    ms = e0.elapsed_time(e1) / replays
            except RuntimeError:
                torch.cuda.synchronize()
            finally:
                self.ring = ring
            for graphs in self._ring_graphs:
                bury(*graphs)
            times[mode] = ms
        # every rank must pick the same placement: the slowest rank's time per mode, ties in the fixed order above
        t = torch.tensor([min(times[m], 1e9) for m in ("graph", "after", "graph-fork")], device=self.x.device, dtype=torch.float64)
        if dist.is_initialized() and dist.get_world_size(self.reducer.group) > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.reducer.group)
        vals = [float(v) for v in t.tolist()]
        best = ("graph", "after", "graph-fork")[min(range(3), key=lambda i: vals[i])]
        self.allreduce_probe = {"ms_per_step": dict(zip(("graph", "after", "graph-fork"), vals)), "chosen": best, "replays": replays}
        self.allreduce, self.in_graph = best, best != "after"
        self._ring_graphs, self._ring_loss, self._ring_outputs = [], [], []
        self._capture(1)

    def _probe_step(self):
        self._replay_graphs(0)
        self.reducer.reduce(collective=not (self.in_graph and self._ring_graphs), replayed=True)

    def __del__(self):
        try:
            from .surface import bury

            for graphs in self.__dict__.get("_ring_graphs", ()):
                bury(*graphs)  # (a hipGraph must not be destroyed while a stream is capturing: surface.py)
        except Exception:  # noqa: BLE001 -- interpreter shutdown
            pass

    def _fast_path(self):
        """forward + loss behind one autograd node (fused_step.SamplerStepFunction / ops.SamplerStepLossFunction): training
        mode with projection, (B,N,3) input, and a batch small enough that the pair scan splits clouds.  The task term is the
        benchmark's stand-in mean(proj) inside the node, or any callable on the projected points OUTSIDE it (its gradient
        enters the node's backward as an explicit tensor; needs the fc4-in-scan, keys-mode step)."""
        net = self.net
        if net.__dict__.get("_sn_sync_bn") is not None:  # statistics over all ranks: collectives between the layers
            return False
        if not self.fused_loss or not net.training or net.skip_projection or net.input_shape != "bnc":
            return False
        if not (getattr(net, "standard_arch", True) or getattr(net, "standard_arch_out_bn", False)) and self.task_loss is not None:
            # the sampler variants of the TF packages (reconstruction: other widths, no BatchNorm in the FC head, clamped temperature;
            # classification: a BatchNorm on the head's output): the single-node LOSS applies to them as it is (the head runs through
            # net._features, the temperature through the module's own clamp); the fc4-in-scan / outside-task forms do not
            return False
        from ._lib import lib

        B, N, _ = self.x.shape
        split = lib.sn_pairscan_colmin_splits(B, N, net.num_out_points) > 1
        if self.task_loss is None:
            # (batches that fill the chip with one workgroup per cloud: same single-node loss behind the plain scan -- fc4 as a
            #  launch of its own, per-cloud loss partials; ops.step_loss_forward)
            return split or N <= 2048
        if not split:
            return False
        if self.task_loss is not None:
            from .fused_step import external_task_supported

            return bool(self.fused_head and getattr(net, "use_hip_mlp", False) and external_task_supported(net, self.x))
        return True

    def _step(self, boundary=None):
        from . import surface

        with surface.suspended():  # the engine places the launches itself: the module surface stays op by op under it
            return self._step_impl(boundary)

    def _step_impl(self, boundary=None):
        """One step (eager, or under capture).  boundary: callable invoked where the FC head's gradients are complete --
        set while capturing the split step; the node then runs without autograd (same thread: a capture may only be ended by
        the thread that began it, and autograd would run the backward on its device thread)."""
        net, x = self.net, self.x
        if self.reducer is not None:
            self.reducer.begin_step()  # this step's backward overwrites the kernel-written gradient views
        if self._fast_path():
            from . import ops

            # (registration architecture, or the same with the classification sampler's BatchNorm on the head's output)
            std = getattr(net, "standard_arch", True) or getattr(net, "standard_arch_out_bn", False)
            T = net.project._temperature
            floor = getattr(net.project, "_temperature_floor", None) is not None
            t_sink = None
            # (floor: the reconstruction variant squares max(T, floor); with a reducer and nothing else that autograd accumulates,
            #  the kernels get the clamped value as data and write its gradient into the bucket as well -- the clamp's gate is then
            #  one in-place launch behind the backward instead of autograd's compare / select / accumulate and a cleared bucket)
            gate = floor and self.reducer is not None and T.requires_grad and boundary is None and self.task_loss is None
            if self.reducer is not None and T.requires_grad and (not floor or gate):
                self.reducer._rebind()  # (after an optimizer.zero_grad(): T.grad is the bucket's view again)
                t_sink = T.grad.view(-1)[:1]  # a view of the flat bucket: written in place, nothing to zero or accumulate
                if self.reducer.autograd_accumulated > 1:
                    # parameters besides the temperature that torch differentiates (a SyncBatchNorm on the head's output): their
                    # gradients are ACCUMULATED into their slices by autograd, step after step unless they are cleared here
                    self.reducer.zero_grad(keep=T)
            elif self.reducer is not None:
                self.reducer.zero_grad()
            weight = self.gamma + self.delta * net.num_out_points
            if boundary is not None:
                from . import fused_step

                loss, y, proj = fused_step.sampler_step_direct(net, x, self.alpha, self.lmbda, weight, t_sink, self._one, boundary)
                self.outputs = (y, proj)
                return loss
            if self.task_loss is not None:
                # main.py:507-531: the task network sits on the projected points.  The node returns a differentiable proj; the
                # task loss's gradient re-enters its backward as an explicit tensor.  The node's loss VALUE is written by the
                # backward's last launch (deferred tail), so the total is formed afterwards.
                from .fused_step import sampler_step

                loss, y, proj = sampler_step(net, x, self.alpha, self.lmbda, weight, t_sink, True, mean_proj=False)
                task = self.task_loss(proj if net.output_shape == "bnc" else proj.permute(0, 2, 1))
                self.outputs = (y.detach(), proj.detach())
                if task.requires_grad:
                    torch.autograd.backward([loss, task], [self._one, self._one.to(task.dtype)])
                else:  # a task term that does not depend on the sampler: the projection branch gets a zero gradient
                    loss.backward(self._one)
                return loss.detach() + task.detach()
            B, N, _ = x.shape
            if gate:
                T = torch.clamp(T.detach(), min=net.project._temperature_floor)
            elif floor:
                T = net.project._t()  # max(T, floor): the kernels square what they are given; its gradient gate is autograd's
            if std and self.fused_head and net.use_hip_mlp and ops.lib.sn_pairscan_colmin_splits(B, N, net.num_out_points) > 1:
                from .fused_step import sampler_step

                loss, y, proj = sampler_step(net, x, self.alpha, self.lmbda, weight, t_sink, True)  # fc4 inside the scan
            else:
                y = net._features(x.permute(0, 2, 1), x)  # (B,3,M)
                loss, proj = ops.SamplerStepLossFunction.apply(y, x, T, net.project._group_size, net.project._min_sigma_f,
                                                               self.alpha, self.lmbda, weight, t_sink, True)
            self.outputs = (y.detach(), proj.detach())  # simplified cloud (B,3,M), projected cloud (B,M,3)
            loss.backward(self._one)  # preallocated upstream gradient: no ones_like fill per step
            if gate:
                net.project._gate_floor_(t_sink)
            return loss.detach()
        if self.reducer is not None:
            self.reducer.zero_grad()
        simp, proj = net(x)
        self.outputs = (simp.detach(), proj.detach())  # in the module's output_shape
        lsimp = net.get_simplification_loss(x, simp, net.num_out_points, self.gamma, self.delta)
        if self.task_loss is None and net.training and not net.skip_projection and getattr(net.project, "_temperature_floor", None) is None:
            # alpha * L_simp + lmbda * sigma + mean(proj) in one fused kernel pair (same value as the composition below; it squares
            # the parameter itself: not for the variant whose sigma is max(T, floor)^2)
            from . import ops

            loss = ops.SamplerLossFunction.apply(lsimp, net.project._temperature, proj, self.alpha, self.lmbda,
                                                 net.project._min_sigma_f)
        else:
            task = self.task_loss(proj) if self.task_loss is not None else proj.mean()
            loss = self.alpha * lsimp + self.lmbda * net.get_projection_loss() + task
        loss.backward(self._one)  # preallocated upstream gradient: no ones_like fill per step
        return loss.detach()

    def _capture_step(self, pool):
        """Captures one step on self.x -> ([graph] | [graph 1, graph 2], loss, outputs)."""
        if not self.split:
            g = torch.cuda.CUDAGraph()
            # thread_local: API calls of other threads (the RCCL watchdog polling events) must not invalidate the capture
            fork = self.in_graph and self.allreduce == "graph-fork" and getattr(self.net, "use_hip_mlp", False)
            prev = getattr(self.net, "_after_fc_grads", None)
            if fork:
                self.reducer.capture_fork = True
                self.net._after_fc_grads = self.reducer._early_ready
            try:
                # (with a collective in play the capture keeps torch's own capture stream: the process group's watchdog thread still
                #  polls the END events of the warm-up's eager collectives, which were recorded on the warm-up stream -- HIP refuses
                #  a query of an event whose stream is capturing, and the watchdog aborts the process)
                kw = {} if (self.reducer is not None and self.reducer.collective) else {"stream": self._cap_stream}
                with torch.cuda.graph(g, pool=pool, capture_error_mode="thread_local", **kw):
                    loss = self._step()
                    if self.in_graph:
                        self.reducer.reduce()  # captured: the collective(s) replay with the step
            finally:
                if fork:
                    self.reducer.capture_fork = False
                    self.net._after_fc_grads = prev
            return [g], loss, self.outputs
        import gc

        graphs = [torch.cuda.CUDAGraph()]

        def boundary():  # FC-head gradients enqueued: close graph 1, open graph 2 on the same stream
            graphs[-1].capture_end()
            graphs.append(torch.cuda.CUDAGraph())
            graphs[-1].capture_begin(pool=pool, capture_error_mode="thread_local")

        torch.cuda.synchronize()
        gc.collect()
        cap = torch.cuda.Stream()
        cap.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(cap), torch.no_grad():
            graphs[0].capture_begin(pool=pool, capture_error_mode="thread_local")
            try:
                loss = self._step(boundary)
            finally:
                graphs[-1].capture_end()
        torch.cuda.current_stream().wait_stream(cap)
        if len(graphs) != 2:
            raise RuntimeError("split capture: the FC / conv boundary was not reached exactly once")
        return graphs, loss, self.outputs

    def _capture(self, warmup):
        if self.reducer is None:
            for p in self.net.parameters():
                p.grad = None
        # (the warm-up passes and the captures share ONE stream: per-stream scratch the warm-up created -- the task network's
        #  counters and constant tables -- is found again under capture instead of being allocated and filled inside the graph)
        side = self._cap_stream = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self._step()
                if self.in_graph:
                    self.reducer.reduce()  # the communicator must exist before a collective can be captured
                if self.reducer is None:
                    for p in self.net.parameters():
                        p.grad = None
        torch.cuda.current_stream().wait_stream(side)
        from .surface import _collect_before_capture

        _collect_before_capture()  # (no graph-holding garbage may be destroyed while the capture below is in progress)
        pool = torch.cuda.graph_pool_handle()
        bufs = self.ring if self.ring is not None else [self.x]
        for buf in bufs:
            if buf.shape != bufs[0].shape or buf.device != bufs[0].device or not buf.is_contiguous():
                raise ValueError("input_ring entries must be contiguous tensors of one shape on one device")
            self.x = buf
            try:
                graphs, loss, outputs = self._capture_step(pool)
            except RuntimeError as e:
                if not self.in_graph or self._ring_graphs:
                    raise
                # the collective could not be captured on this stack: one collective from Python behind every replay instead
                import warnings

                warnings.warn("SamplerTrainStep: capturing the gradient all-reduce inside the step graph failed (%s); falling "
                              "back to allreduce='after'" % str(e).splitlines()[0])
                torch.cuda.synchronize()
                self.in_graph, self.allreduce = False, "after"
                graphs, loss, outputs = self._capture_step(pool)
            self._ring_graphs.append(graphs)
            self._ring_loss.append(loss)
            self._ring_outputs.append(outputs)
        self.x = bufs[0]
        self.graph = self._ring_graphs[0][0]
        self.loss = self._ring_loss[0]
        # the graphs hold raw pointers into the module's persistent forward buffers (pointnet._ForwardPlan): keep them alive
        # for as long as the graphs, whatever happens to the module's own list
        self._plan_refs = list(self.net.__dict__.get("_sn_plans", ()))

    def _replay_graphs(self, i):
        graphs = self._ring_graphs[i]
        graphs[0].replay()
        if len(graphs) == 2:
            self.reducer.reduce_early_async()  # FC-head segment: all-reduce on the side stream beside graph 2
            graphs[1].replay()
        self.loss, self.outputs = self._ring_loss[i], self._ring_outputs[i]

    def replay(self, i):
        """One step on ring entry i (whatever the caller wrote into input_ring[i]); returns the (static) loss tensor."""
        if self.ring is None:
            raise RuntimeError("replay(i) needs an input ring; call the step with a batch instead")
        if self._ring_graphs:
            self._replay_graphs(i)
        else:
            self.x = self.ring[i]
            self.loss = self._step()
        if self.reducer is not None:
            self.reducer.reduce(collective=not (self.in_graph and self._ring_graphs), replayed=bool(self._ring_graphs))
        return self.loss

    def check(self):
        """Host-side health check of the step's device-side launch state (the FC chain kernels' hand-off error words): raises
        SampleNetHipError when a step since the last check ran on incomplete data (its loss was NaN-poisoned on the device
        already).  Synchronises the device: call it where the loss is read back, not per step."""
        from . import pointnet

        if getattr(self.net, "use_hip_mlp", False):
            pointnet.check_chain_errors(self.net)

    def __call__(self, x):
        """Runs one step on batch x (same shape as example_x); returns the (static) loss tensor.
        Gradients are in p.grad afterwards (cross-rank averaged when a reducer is attached)."""
        if self.ring is not None:
            raise RuntimeError("this step was built on an input ring: fill input_ring[i] in place and call replay(i)")
        self.x.copy_(x, non_blocking=True)
        if self._ring_graphs:
            self._replay_graphs(0)
        else:
            self.loss = self._step()
        if self.reducer is not None:
            self.reducer.reduce(collective=not (self.in_graph and self._ring_graphs), replayed=bool(self._ring_graphs))
        return self.loss
