"""Data-parallel training step: one process per MI355X, RCCL (torch.distributed "nccl") over xGMI.

What the reference does with ``DistributedDataParallel`` + ``torch.optim.Adam``
(utils/main_utils.py:105-117, :250-261; main-avid.py:155-180) is done here with

* ``FlatParams``   — every parameter (and its gradient) is a strided view into ONE flat fp32 buffer,
                     laid out in reverse registration order so buckets complete front-to-back while
                     backward runs (conv5x / heads first).  Views keep each tensor's memory format,
                     so the channels-last conv weights stay channels-last.
* ``GradBuckets``  — per-bucket async all-reduce (sum) launched from post-accumulate-grad hooks as soon
                     as the bucket's last gradient is written, i.e. overlapped with the rest of
                     backward; ``finish()`` makes the compute stream wait.  The 1/world average is
                     folded into the optimizer kernel's ``grad_scale``.
* ``FlatBuffers``  — the BatchNorm running statistics as views of one flat buffer: DDP's buffer broadcast
                     (``broadcast_buffers=True``, the default of utils/main_utils.py:112) is ONE 78 KB collective.
* ``TrainStep``    — fwd -> criterion -> bwd (+ overlapped all-reduce) -> one fused flat-Adam launch;
                     ``state_dict`` / ``load_state_dict`` in torch.optim.Adam's format (the reference's
                     CheckpointManager saves ``optimizer.state_dict()``, main-avid.py:115,127,138).

The comm layer is backend-agnostic (gloo on CPU in tests/test_distributed_cpu.py); only the Adam
kernel needs the GPU.  BatchNorm statistics stay per-rank — the reference has no SyncBN.
"""
from __future__ import annotations

import os
import weakref
import torch
import torch.distributed as dist


_FLAT_OF = {}           # id(parameter) -> weakref(FlatParams it is a view of)


def flat_of(params):
    """The FlatParams whose parameters are exactly ``params`` (any order), or None."""
    params = [p for p in params if p.requires_grad]
    if not params:
        return None
    ref = _FLAT_OF.get(id(params[0]))
    flat = ref() if ref is not None else None
    if flat is None or len(flat.params) != len(params):
        return None
    mine = {id(p) for p in flat.params}
    if any(id(p) not in mine for p in params) or any(p.data_ptr() != flat.flat.data_ptr() + 4 * o
                                                     for p, o in zip(flat.params, flat.offsets)):
        return None
    return flat


def _dist_on():
    """Collectives are in play: more than one rank (or AVID_FORCE_DIST=1, which drives the RCCL path on a
    single-rank group so it can be exercised on a one-GPU box)."""
    import os
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or os.environ.get("AVID_FORCE_DIST", "0") == "1"


class FlatParams:
    """Re-seat ``module``'s parameters and gradients as views of two flat buffers."""

    def __init__(self, module):
        """``module``: an nn.Module, or the parameters themselves (an optimizer's: Adam below)."""
        src = module.parameters() if isinstance(module, torch.nn.Module) else module
        params = [p for p in src if p.requires_grad]
        self.params = list(reversed(params))            # backward produces gradients roughly in this order
        dev, dt = self.params[0].device, self.params[0].dtype
        self.offsets, off = [], 0
        for p in self.params:
            if not (p.is_contiguous() or p.movedim(1, -1).is_contiguous()):
                raise ValueError("FlatParams needs dense parameters")
            self.offsets.append(off)
            off += (p.numel() + 3) // 4 * 4             # keep every slice 16-byte aligned
        self.numel = off
        self.flat = torch.zeros(off, dtype=dt, device=dev)
        self.grad = torch.zeros(off, dtype=dt, device=dev)
        self.grad_views = []
        for p, o in zip(self.params, self.offsets):
            view = self.flat[o:o + p.numel()].as_strided(p.shape, p.stride())
            view.copy_(p.data)
            p.data = view
            p.grad = self.grad[o:o + p.numel()].as_strided(p.shape, p.stride())
            self.grad_views.append(p.grad)
            _FLAT_OF[id(p)] = weakref.ref(self)

    def seat_grads(self):
        """`.grad` of every parameter is its view of the flat gradient buffer again (after a zero_grad(set_to_none=True):
        the kernels wrote the buffer, not the attribute)."""
        for p, v in zip(self.params, self.grad_views):
            if p.grad is not v:
                p.grad = v

    def zero_grad(self):
        self.grad.zero_()
        for p, o in zip(self.params, self.offsets):     # re-seat if something replaced .grad (set_to_none)
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * o:
                p.grad = self.grad_views[self.offsets.index(o)]


class FlatBuffers:
    """The floating-point buffers of ``module`` (BatchNorm running_mean / running_var) re-seated as views of one
    flat tensor, so that broadcasting rank 0's buffers — what DistributedDataParallel(broadcast_buffers=True)
    does before every forward — is a single small collective.  ``num_batches_tracked`` (int64) advances by one
    per step on every rank and is identical everywhere by construction."""

    def __init__(self, module: torch.nn.Module):
        # ONE owner.  models.AV_Wrapper keeps its floating-point buffers in its own flat tensor (`_bn_flat`, what torch's DDP
        # broadcasts) and re-seats them there on every forward: a second flat tensor here would be orphaned by the first
        # evaluation / per-layer forward, and sync_buffers() would then broadcast a tensor nobody reads.  So the module's
        # buffer is ADOPTED (looked up at every use: `.to()` / `.cuda()` re-create it), and only a module without one gets
        # a flat tensor of this class's making.
        self.module = module if (hasattr(module, "_seat_flat_buffers") and getattr(module, "_bn_flat", None) is not None) else None
        if self.module is not None:
            module._seat_flat_buffers()
            mods = dict(module.named_modules())
            self.bufs, self.offsets = [], []
            for n, off, cnt in module._bn_layout:
                owner, _, leaf = n.rpartition(".")
                self.bufs.append(getattr(mods[owner], leaf))
                self.offsets.append(off)
            self.numel = module._bn_flat.numel()
            self._flat = None
            return
        self.bufs = [b for b in module.buffers() if b.is_floating_point()]
        self.offsets, off = [], 0
        for b in self.bufs:
            self.offsets.append(off)
            off += (b.numel() + 3) // 4 * 4
        self.numel = off
        if not self.bufs:
            self._flat = None
            return
        self._flat = torch.zeros(off, dtype=self.bufs[0].dtype, device=self.bufs[0].device)
        for b, o in zip(self.bufs, self.offsets):
            view = self._flat[o:o + b.numel()].view(b.shape)
            view.copy_(b)
            b.data = view

    @property
    def flat(self):
        if self.module is not None:
            self.module._seat_flat_buffers()             # (no-op unless the buffers were re-created since)
            return self.module._bn_flat
        return self._flat

    def broadcast(self, src=0, async_op=False):
        if self.flat is None or not _dist_on():
            return None
        return dist.broadcast(self.flat, src, async_op=async_op)


def lib_timing():
    from . import lib
    return lib.TIMING


class GradBuckets:
    """Bucketed, backward-overlapped gradient all-reduce over a ``FlatParams`` gradient buffer."""

    def __init__(self, flat: FlatParams, bucket_bytes: int = 16 << 20, average: bool = False, hooks=None):
        """``average``: the collective leaves the MEAN over ranks in the buffer (what torch's DDP hands the optimizer) instead
        of the sum (TrainStep folds 1 / world into its Adam launch).  ``hooks``: autograd hooks on every parameter — default:
        only where no kernel reports its gradients itself (CPU tensors)."""
        self.flat = flat
        self.average = average
        self.world = dist.get_world_size() if _dist_on() else 1
        self.comm = _dist_on()
        self.bounds, self.bucket_of = [], []
        start, cur = 0, 0
        for i, (p, o) in enumerate(zip(flat.params, flat.offsets)):
            end = o + (p.numel() + 3) // 4 * 4
            self.bucket_of.append(len(self.bounds))
            if (end - start) * 4 >= bucket_bytes or i == len(flat.params) - 1:
                self.bounds.append((start, end))
                start = end
        self.counts = [0] * len(self.bounds)
        for b in self.bucket_of:
            self.counts[b] += 1
        self.pending = list(self.counts)
        self.launched = [False] * len(self.bounds)        # bucket b's collective has been issued this step
        self.reported = [False] * len(flat.params)        # parameter i's gradient was reported (ready(i)) this step
        self.works = []
        self.hooks = []
        self.comm_used = []           # every collectives' stream a bucket went out on this step (normally one)
        self.step_set = None          # the step's StreamSet, resolved once per step (first bucket)
        self.measure = False          # bench.py: event-time the compute stream's wait for the collectives in finish()
        self.on_autograd = None       # called from every autograd hook (DistributedDataParallel: arm the end-of-backward callback)
        self.wait_events = []
        self.producers = [set() for _ in self.bounds]     # streams that issued gradients of each bucket
        # Autograd hooks only where the kernels do not report their gradients themselves (ops.GradSlots, the
        # CUDA path): 141 tensor hooks cost ~0.4 ms of backward per step on the single-rank check.  A gradient
        # that arrives through autograd anyway (a parameter used twice) is picked up by finish().
        if (self.comm and not flat.grad.is_cuda) if hooks is None else hooks:
            for i, p in enumerate(flat.params):
                self.hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(i)))

    def ready(self, i, stream=None):
        """Gradient of parameter i has been issued on ``stream`` (default: the current one) — autograd hook,
        ops.GradSlots for kernels that write the flat buffer directly, or a launch program's segment end: launch its
        bucket's all-reduce once the bucket is complete."""
        self.reported[i] = True
        if not self.comm:
            return
        b = self.bucket_of[i]
        if self.flat.grad.is_cuda:
            self.producers[b].add(stream if stream is not None else torch.cuda.current_stream(self.flat.grad.device))
        self.pending[b] -= 1
        if self.pending[b] == 0 and not self._capturing():
            self._launch(b)

    def _capturing(self):
        return self.flat.grad.is_cuda and torch.cuda.is_current_stream_capturing()

    def _make_hook(self, i):
        view = self.flat.grad_views[i]

        def hook(param):
            # a gradient that came through autograd into a tensor of its own (`.grad` was None: zero_grad(set_to_none=True))
            # belongs in the flat buffer the collectives and the optimizer read
            g = param.grad
            if g is not None and g.data_ptr() != view.data_ptr():
                view.copy_(g)
                param.grad = view
            if self.on_autograd is not None:
                self.on_autograd()
            self.ready(i)
        return hook

    def _launch(self, b):
        self.launched[b] = True
        self._issue(b)

    def _issue(self, b):
        """Issue bucket b's all-reduce on the collectives' stream of the step's StreamSet (avid_hip/streams.py: a stream
        on a dispatch pipe of its own), ordered behind every stream that produced one of the bucket's gradients — the
        compute streams never wait here, and nothing waits for the collective before ``finish()``.  The collective is
        issued as a stream-synchronous op under that stream: torch's NCCL process group then runs it ON that stream
        (c10d: ``asyncOp = false`` uses the current stream) instead of its internal one, whose hardware queue we could
        not choose — two queues on one dispatch pipe serialise, and a queue waiting for an event of its pipe-mate
        stalls both (the 6-8 ms a one-rank group used to add)."""
        s, e = self.bounds[b]
        if not self.flat.grad.is_cuda:
            # (gloo has no AVG: the mean is taken in finish(), behind the wait)
            self.works.append(dist.all_reduce(self.flat.grad[s:e], async_op=True))
            return
        from . import lib, streams
        dev = self.flat.grad.device
        # ONE StreamSet per step: the first bucket resolves it (from whichever of the set's streams is current — the
        # per-layer path reports deferred weight gradients with the trailing stream current), the rest of the step reuses
        # it, so every bucket goes out on the same comm stream of the same RCCL communicator
        ss = self.step_set
        if ss is None:
            ss = self.step_set = streams.current_set(dev)
        prod = self.producers[b] or {ss.main, ss.side, ss.trail}      # unknown producers: everything
        cs = ss.comm
        for st in prod:
            lib.call("avid_stream_wait", cs.cuda_stream, st.cuda_stream)
        native = self.average and dist.get_backend() == "nccl"       # (RCCL averages; gloo — the two-ranks-on-one-GPU tests — cannot)
        with torch.cuda.stream(cs):
            dist.all_reduce(self.flat.grad[s:e], op=dist.ReduceOp.AVG if native else dist.ReduceOp.SUM)
            if self.average and not native and self.world > 1:
                self.flat.grad[s:e].div_(self.world)
        if all(cs is not c for c in self.comm_used):
            self.comm_used.append(cs)

    def exposed_wait_ms(self):
        """Mean time per step the compute stream spent waiting for the gradient collectives in ``finish()`` since
        ``measure`` was switched on (synchronises; clears the record)."""
        if not self.wait_events:
            return None
        torch.cuda.synchronize()
        ms = [a.elapsed_time(b) for a, b in self.wait_events]
        self.wait_events = []
        return sum(ms) / len(ms)

    def finish(self):
        """Launch whatever did not fire (unused parameters, gradients that came through autograd) and make the
        current stream wait for all buckets."""
        if self.comm and self._capturing():
            # Inside a hipGraph capture the bucketed collectives, issued while the backward pass is being recorded, turn
            # into joins of the streams in the graph (round 2: 23.8 ms per replay against 14 ms eager).  A captured step
            # reduces the whole gradient buffer with ONE collective behind the backward pass instead: the all-reduce is
            # exposed (85 MB), the rest of the graph keeps its shape.
            from . import ops
            dev = self.flat.grad.device
            cur = torch.cuda.current_stream(dev)
            cur.wait_stream(ops.side_stream(dev, 1))         # the audio tower's gradients
            native = self.average and dist.get_backend() == "nccl"
            dist.all_reduce(self.flat.grad, op=dist.ReduceOp.AVG if native else dist.ReduceOp.SUM)
            if self.average and not native and self.world > 1:
                self.flat.grad.div_(self.world)
            self.launched = [True] * len(self.bounds)
        if self.comm:
            for b, left in enumerate(self.pending):
                if self.launched[b]:
                    continue
                if left > 0:
                    self.producers[b].clear()        # not all producers are known: wait for every stream
                self._launch(b)
            timed = self.measure and self.flat.grad.is_cuda and not self._capturing()
            if timed:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            for w in self.works:
                w.wait()                             # (CPU tensors: gloo)
            if self.works and self.average and self.world > 1:
                self.flat.grad.div_(self.world)
            for cs in self.comm_used:
                torch.cuda.current_stream(self.flat.grad.device).wait_stream(cs)
            self.comm_used = []
            if timed:
                e1.record()
                self.wait_events.append((e0, e1))
        self.reset()

    def reset(self):
        """The per-step state as a step finds it.  ``finish()`` ends with it; DistributedDataParallel.forward starts with it, so a
        backward pass that raised half-way (out of memory, 'backward a second time') cannot leave counts behind that keep the
        next step's buckets from ever launching — torch's reducer does the same in prepare_for_backward."""
        self.works = []
        self.comm_used = []
        self.step_set = None
        self.pending = list(self.counts)
        self.launched = [False] * len(self.bounds)
        self.reported = [False] * len(self.flat.params)
        for p in self.producers:
            p.clear()


class TrainStep:
    """One AVID training step (main-avid.py:155-180) on this rank's GPU.

    ``step(video, audio, index)`` returns the (device) loss tensor; nothing synchronises the host.
    Adam hyper-parameters follow the shipped configs (lr 2e-4, betas (0.9, 0.999), L2 wd 1e-5).
    """

    def __init__(self, model, criterion, lr=2e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-5,
                 bucket_bytes=16 << 20, broadcast_buffers="step", _wrapper=False):
        """``broadcast_buffers``: DistributedDataParallel broadcasts rank 0's BatchNorm buffers before EVERY
        forward (utils/main_utils.py:112, default ``broadcast_buffers=True``).  A training-mode forward never
        reads them (it normalises with the batch statistics), and rank 0's own buffers are never overwritten, so
        the only observable effect is what a non-zero rank evaluates / saves with.  ``"step"`` (default) reproduces
        DDP literally with ONE flat 78 KB broadcast per step (the running statistics are views of one buffer);
        ``"lazy"`` broadcasts at ``sync_buffers()`` only — the caller must invoke it before evaluating or saving on a
        rank other than 0 — ``"off"`` never broadcasts."""
        import os
        self.model, self.criterion = model, criterion
        self.lr, self.betas, self.eps, self.wd = lr, betas, eps, weight_decay
        if broadcast_buffers not in ("lazy", "step", "off"):
            raise ValueError("broadcast_buffers must be 'lazy', 'step' or 'off'")
        self.broadcast_buffers = broadcast_buffers
        if _dist_on():                                   # DDP's construction-time parameter broadcast (C3)
            for p in model.parameters():
                dist.broadcast(p.data, 0)
            for b in model.buffers():
                dist.broadcast(b.data, 0)
        self.flat = FlatParams(model)
        self.flat_buffers = FlatBuffers(model)
        # The audio tower runs on a side stream under the video tower (models/av_wrapper.py): on by default
        # here, because GradBuckets below orders every bucket's collective after the streams that produced it.
        if hasattr(model, "overlap_towers") and os.environ.get("AVID_OVERLAP_TOWERS", "1") == "1":
            model.overlap_towers = True
        if os.environ.get("AVID_BUCKET_MB"):             # tuning knob: gradient all-reduce bucket size
            bucket_bytes = int(float(os.environ["AVID_BUCKET_MB"]) * (1 << 20))
        # (_wrapper: the engine inside DistributedDataParallel below — the optimizer is somebody else's, so the collectives
        #  average, every parameter carries an autograd hook for gradients that do not come out of a launch program, and
        #  there is no Adam state here)
        self.buckets = GradBuckets(self.flat, bucket_bytes, average=_wrapper, hooks=True if _wrapper else None)
        self.m = None if _wrapper else torch.zeros_like(self.flat.flat)
        self.v = None if _wrapper else torch.zeros_like(self.flat.flat)
        self.t = 0
        self.t_dev = torch.zeros((), dtype=torch.int64, device=self.flat.flat.device) \
            if self.flat.flat.is_cuda else None
        # the learning rate lives in device memory as well: a captured graph freezes by-value arguments
        self.lr_dev = torch.full((), float(lr), dtype=torch.float32, device=self.flat.flat.device) \
            if self.flat.flat.is_cuda else None
        self.graph = None
        self.twt = self.slots = None
        if self.flat.flat.is_cuda:
            from . import ops
            self.twt = ops.TransposedWeights(self.flat.params)   # dgrad weight repack: one launch per step
            # backward kernels write parameter gradients straight into the flat buffer (no per-parameter add)
            self.slots = ops.GradSlots(self.flat.params, self.flat.grad_views, on_ready=self.buckets.ready)

    def set_lr(self, lr):
        """Change the learning rate (an LR scheduler's hook; also reaches a captured graph)."""
        self.lr = float(lr)
        if self.lr_dev is not None:
            self.lr_dev.fill_(self.lr)

    def sync_buffers(self):
        """Every rank takes rank 0's BatchNorm running statistics (see ``broadcast_buffers``): call before
        evaluating the model or saving it on a rank other than 0."""
        self.flat_buffers.broadcast(0)

    def _plan_backward(self, pl, fa, video, audio, dv, da):
        """The backward launch program of the model (called from its autograd node): in one piece, or — with gradient
        collectives — cut where a bucket becomes complete, so that its all-reduce starts under the rest of the pass."""
        self._step_plan = pl
        if not self.buckets.comm or self.buckets._capturing():
            pl.backward(fa, video, audio, dv, da, self.flat.grad)
            return
        ba, begin = None, 0
        for end, ready in pl.segments(self.buckets.bucket_of, self.buckets.counts):
            ba = pl.backward(fa, video, audio, dv, da, self.flat.grad, begin, end, ba)
            for i, st in ready:
                self.buckets.ready(i, pl.stream_objs[st])
            begin = end

    def _forward_backward_plan(self, video, audio, index):
        """The step through the compiled launch programs (avid_hip/plan.py); None if the model is outside them."""
        from . import plan
        if not self.flat.grad.is_cuda:
            return None
        with plan.engine(self):
            out = plan.run(self.model, video, audio)
            if out is None:
                return None
            loss, _ = self.criterion(out[0], out[1], index)
            loss.backward()
        self.buckets.finish()
        return loss

    def forward_backward(self, video, audio, index):
        if self.broadcast_buffers == "step":
            self.sync_buffers()
        loss = self._forward_backward_plan(video, audio, index)
        if loss is not None:
            return loss
        from . import ops
        helper = None
        if self.twt is not None:
            self.twt.prepare_wino()
            self.twt.uready = False
        if self.twt is not None and self.flat.grad.is_cuda and ops.DEFER_WGRAD and self._defer_ok() and not lib_timing():
            # What the backward needs but the forward does not — zeroed gradients, the transposed weight copies, the
            # Winograd transforms of the weights — runs on a helper stream next to the forward instead of in front of /
            # behind it (the weights cannot change in between: this method owns the step).
            # (the helper is the weight-gradient stream, idle during the forward: a FIFTH stream would share one of
            # the runtime's four hardware queues with a busy one and serialise behind it — 4700 -> 3100 clips/s)
            # How this shares the chip with the stem convolution, the forward's first 0.87 ms, decides what it costs
            # (tools/trace_timeline.py on rocprofv3 traces, tools/host_lead.py for the phases):
            #  * issued like this, in front of the model, the gradient fill starts together with the stem kernel, whose
            #    persistent workgroups hold every CU: the fill starves until the stem ends (795 us instead of 22) and the
            #    rest follows in the ~130 us behind it — the stem pays 25 us, the side work nothing else;
            #  * anything that DELAYS the stem's start by ~25 us (a transform launch in front of it on the main stream)
            #    lets the side work win that race: it then runs inside the stem's first 100-150 us and the stem takes
            #    1.35-1.47 ms instead of 0.87 (the forward 4.8 instead of 4.25 ms) — round 3's unexplained "slower with
            #    the transforms hoisted"; a 96 MB fill or copy in front of the model does the same;
            #  * started strictly BEHIND the stem (an event behind its launch) it shares the chip with the pooling pass
            #    and conv2x instead: +0.07 ms.
            # The co-runner that does the damage is the weight transposer, not the fill (tools/corun.py: the stem alone
            # 0.87-0.97 ms; with an 85 MB fill released at the same moment or 30 us earlier 0.90-0.94; with the transposes
            # 1.38-1.44 — their tens of thousands of short 4-wave workgroups leave the stem's persistent waves unevenly
            # spread over the SIMDs, and the slowest workgroup sets the kernel's time; released 30 us later: 0.89).
            # So nothing goes in front of the model on the main stream, and the fill goes first on the helper: it starts
            # with the stem, starves, and keeps everything behind it away from the stem's start.  (Making that explicit —
            # the transposes behind an event recorded after the stem convolution — was measured too: the event's latency
            # moves them from the pooling pass into conv2x, forward 4.25 -> 4.44 ms.)
            cur, helper = ops.wgrad_stream(self.flat.grad.device)
            helper.wait_stream(cur)
            with torch.cuda.stream(helper):
                self.flat.zero_grad()
                self.twt.refresh_wino(False)
                self.twt.refresh()
                self.twt.refresh_wino(True)
        else:
            self.flat.zero_grad()
            if self.twt is not None:
                self.twt.refresh_wino(False)
        if self.twt is not None:
            with self.twt.armed_wino():
                video_emb, audio_emb = self.model(video, audio)
        else:
            video_emb, audio_emb = self.model(video, audio)
        loss, _ = self.criterion(video_emb, audio_emb, index)
        if self.twt is not None:
            if helper is not None:
                torch.cuda.current_stream().wait_stream(helper)
            else:
                self.twt.refresh()                   # after the forward: whatever the weights are now
                self.twt.refresh_wino(True)
            with self.twt.armed(), self.twt.armed_wino(), self.slots.armed(), \
                    ops.deferred_wgrads(enabled=self._defer_ok()):
                loss.backward()
        else:
            loss.backward()
        self.buckets.finish()
        return loss

    def _defer_ok(self):
        """Trailing weight-gradient stream: always (with or without gradient collectives — the collectives' stream is
        placed on a dispatch pipe of its own, avid_hip/streams.py; a captured step decides for itself in ops)."""
        return True

    def optimizer_step(self):
        from . import ops
        self.t += 1
        ops.adam_flat(self.flat.flat, self.flat.grad, self.m, self.v, self.lr, self.betas[0], self.betas[1],
                      self.eps, self.wd, self.t, grad_scale=1.0 / self.buckets.world, step_dev=self.t_dev,
                      lr_dev=self.lr_dev)

    def _optimizer_step_overlapped(self, pl):
        """The same update in two launches: every parameter but the video stem's three on the fourth stream, which the
        backward program made wait for exactly their gradients (plan.Plan: ``adam_early``) — it runs beside the stem's
        weight gradient, the step's last kernel — then the stem's on the compute stream.  Same arithmetic per element."""
        from . import ops
        self.t += 1
        n0 = pl.adam_early
        f, g, m, v = self.flat.flat, self.flat.grad, self.m, self.v
        kw = dict(grad_scale=1.0 / self.buckets.world, step_dev=self.t_dev, lr_dev=self.lr_dev)
        main, fourth = torch.cuda.current_stream(), pl.stream_objs[3]
        with torch.cuda.stream(fourth):
            ops.adam_flat(f[:n0], g[:n0], m[:n0], v[:n0], self.lr, self.betas[0], self.betas[1], self.eps, self.wd, self.t, **kw)
        main.wait_stream(fourth)
        ops.adam_flat(f[n0:], g[n0:], m[n0:], v[n0:], self.lr, self.betas[0], self.betas[1], self.eps, self.wd, self.t,
                      advance=False, **kw)

    # ---- optimizer / sampler state in torch.optim.Adam's format (main-avid.py:115,127,138 save and restore
    # ``optimizer.state_dict()``; utils/main_utils.py:250-261 builds Adam over model.parameters())
    def _param_order(self):
        """[(index in ``list(model.parameters())`` — frozen parameters included, as torch.optim.Adam built over
        ``model.parameters()`` numbers them (utils/main_utils.py:250-261) —, slot in the flat layout, parameter)]
        for the trainable parameters."""
        slot = {id(p): i for i, p in enumerate(self.flat.params)}
        return [(k, slot[id(p)], p) for k, p in enumerate(self.model.parameters()) if id(p) in slot]

    def _slice(self, flat_tensor, i):
        p, o = self.flat.params[i], self.flat.offsets[i]
        return flat_tensor[o:o + p.numel()].as_strided(p.shape, p.stride())

    def state_dict(self):
        order = self._param_order()
        step = float(int(self.t_dev) if self.t_dev is not None else self.t)
        state = {}
        if step > 0:
            for k, i, _ in order:                  # (a frozen parameter has an index but no state, as in torch)
                state[k] = {"step": torch.tensor(step), "exp_avg": self._slice(self.m, i).detach().clone(),
                            "exp_avg_sq": self._slice(self.v, i).detach().clone()}
        nparams = sum(1 for _ in self.model.parameters())
        sd = {"state": state,
              "param_groups": [{"lr": self.lr, "betas": tuple(self.betas), "eps": self.eps, "weight_decay": self.wd,
                                "amsgrad": False, "maximize": False, "foreach": None, "capturable": False,
                                "differentiable": False, "fused": None, "decoupled_weight_decay": False,
                                "params": list(range(nparams))}]}
        mult = getattr(getattr(self.criterion, "nce_average", None), "multinomial", None)
        if mult is not None:       # the negative sampler's stream position (not part of the reference's checkpoint:
            off = int(mult.offset_dev) if getattr(mult, "offset_dev", None) is not None else int(mult.offset)
            sd["avid_sampler"] = {"seed": int(mult.seed), "offset": off}   # torch's global RNG is not saved either)
        return sd

    def load_state_dict(self, sd):
        order = self._param_order()
        g = sd["param_groups"][0]
        self.betas, self.eps, self.wd = tuple(g["betas"]), g["eps"], g["weight_decay"]
        self.set_lr(g["lr"])
        self.m.zero_()
        self.v.zero_()
        step = 0
        for k, i, _ in order:
            st = sd["state"].get(k, sd["state"].get(str(k)))
            if st is None:
                continue
            self._slice(self.m, i).copy_(st["exp_avg"])
            self._slice(self.v, i).copy_(st["exp_avg_sq"])
            step = max(step, int(float(st["step"])))
        self.t = step
        if self.t_dev is not None:
            self.t_dev.fill_(step)
        mult = getattr(getattr(self.criterion, "nce_average", None), "multinomial", None)
        if mult is not None and "avid_sampler" in sd:
            mult.reseed(sd["avid_sampler"]["seed"], sd["avid_sampler"]["offset"])

    def _poll_errors(self):
        """Out-of-range sample ids raise here (non-blocking look at the device error word, ops.DeviceErrors): the
        criterion's own polls do not run when a captured graph is replayed.  The error surfaces about one step after
        the offending kernels, i.e. after Adam has applied that step — ``ops.check_device_errors()`` is the blocking
        form for checkpoint time."""
        if self.flat.flat.is_cuda:
            from . import ops
            ops.poll_device_errors(self.flat.flat.device)

    def step(self, video, audio, index):
        self._step_plan = None
        loss = self.forward_backward(video, audio, index)
        pl = self._step_plan
        if pl is not None and pl.adam_early and not self.buckets.comm and not lib_timing():
            self._optimizer_step_overlapped(pl)
        else:
            self.optimizer_step()
        self._poll_errors()
        return loss.detach()

    # ---- whole-step hipGraph: ~600 launches replayed with one host call (kills the Python launch overhead)
    def capture(self, video, audio, index):
        """Capture one full step into a hipGraph.  Call after >= 2 eager warm-up steps (Z fixed, workspaces
        sized).  Inputs are copied into static buffers on every ``replay``."""
        self._sv, self._sa, self._si = video.clone(), audio.clone(), index.clone()
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        t_host = self.t
        mult = getattr(getattr(self.criterion, "nce_average", None), "multinomial", None)
        off_host = mult.offset if mult is not None else None
        # (thread-local capture mode: the NCCL process group's watchdog thread polls the events of earlier collectives with
        #  hipEventQuery — under the default global mode that call is illegal while ANY thread captures and aborts the process)
        with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
            self._sloss = self.step(self._sv, self._sa, self._si)
        if self.twt is not None:
            self.twt.frozen = True                   # (the graph holds the table's addresses: ops.prepare_wino)
        self.t = t_host                              # the capture executed nothing: host counters stay where the
        if mult is not None:                         # device counters are
            mult.offset = off_host
        return self

    def replay(self, video=None, audio=None, index=None):
        if video is not None and video.data_ptr() != self._sv.data_ptr():
            self._sv.copy_(video, non_blocking=True)
        if audio is not None and audio.data_ptr() != self._sa.data_ptr():
            self._sa.copy_(audio, non_blocking=True)
        if index is not None:
            self._si.copy_(index, non_blocking=True)
        self.graph.replay()
        self._poll_errors()
        self.t += 1
        mult = getattr(getattr(self.criterion, "nce_average", None), "multinomial", None)
        if mult is not None:
            mult.offset += 1
        return self._sloss


# ---------------------------------------------------------------------------------------------------------------------
# The reference's OWN loop (main-avid.py:155-180: model(...); criterion(...); loss.item(); optimizer.zero_grad(); loss.backward();
# optimizer.step()) on the engine's pieces: the two objects its factories build — utils/main_utils.py:112
# `DistributedDataParallel(model, device_ids=[gpu])` and :250 `torch.optim.Adam(params, lr, weight_decay, betas)` — with the same
# constructors and the same behaviour towards that loop.  Swapping the two names in main_utils.py is the whole change.
class _WrapperEngine(TrainStep):
    """TrainStep's flat buffers / buckets / launch-program hand-over without its criterion and optimizer: what runs under
    `DistributedDataParallel.forward` and the `loss.backward()` the caller issues later."""

    def __init__(self, model, bucket_bytes, broadcast_buffers):
        super().__init__(model, None, bucket_bytes=bucket_bytes, broadcast_buffers="step" if broadcast_buffers else "off",
                         _wrapper=True)
        self._armed = False
        self.buckets.on_autograd = self._arm

    def _arm(self):
        """Once per backward pass: when autograd has run every node, finish the collectives (the caller's stream waits for
        them) and give every parameter its `.grad` — torch's reducer queues the same callback (reducer.cpp: finalize_backward)."""
        if not self._armed:
            self._armed = True
            torch.autograd.Variable._execution_engine.queue_callback(self._after_backward)

    def _after_backward(self):
        self._armed = False
        reported = list(self.buckets.reported)
        self.buckets.finish()
        self._seat_reported(reported)

    def _seat_reported(self, reported):
        """`.grad` = the flat view for every parameter whose gradient was produced this step.  A parameter nothing reported
        (frozen for this step, unused) keeps `.grad is None` if the caller's zero_grad(set_to_none=True) left it so — torch's DDP +
        optimizer then skip it, and handing it the view would hand it the previous step's gradient — unless the buffer was
        zeroed (avid_hip.parallel.Adam.zero_grad: the view is then a true zero gradient and stays seated)."""
        flat = self.flat
        if all(reported) or not any(reported):       # (launch programs report per segment: all of them; CPU / hook-less paths: none)
            flat.seat_grads()
            return
        for p, v, r in zip(flat.params, flat.grad_views, reported):
            if r:
                if p.grad is not v:
                    p.grad = v
            elif p.grad is None:
                v.zero_()                            # stale bytes must not reach an optimizer that reads the flat buffer

    def begin_step(self):
        """A new forward in training mode: whatever a failed backward pass left behind is dropped."""
        self._armed = False
        self.buckets.reset()

    def _plan_backward(self, pl, fa, video, audio, dv, da):
        self._arm()
        super()._plan_backward(pl, fa, video, audio, dv, da)


class DistributedDataParallel(torch.nn.Module):
    """`torch.nn.parallel.DistributedDataParallel(module, device_ids=[gpu])` as utils/main_utils.py:112 builds it, for one
    process per GPU: `.module`, `module.`-prefixed state_dict keys (main-avid.py:100, CheckpointManager), rank 0's parameters and
    buffers broadcast at construction, rank 0's buffers before every training forward (`broadcast_buffers=True`: ONE 78 KB
    collective), and after `loss.backward()` every `p.grad` holds the mean over ranks.  Underneath: the parameters and
    gradients are views of flat buffers, the backward launch program writes the gradients in place, the buckets' all-reduces
    (RCCL, mean) start under the rest of the backward pass on the collectives' stream — no per-parameter copy kernel in either
    direction (torch's reducer: 141 `mul_out` + 141 copies back per step, 1.1 ms of the 12.3 ms step on one rank).

    Not reproduced: `no_sync()` / gradient accumulation over several backward passes, `find_unused_parameters`, comm hooks —
    the reference uses none of them; asking for one raises."""

    def __init__(self, module, device_ids=None, output_device=None, dim=0, broadcast_buffers=True, process_group=None,
                 bucket_cap_mb=None, find_unused_parameters=False, gradient_as_bucket_view=False, **unsupported):
        super().__init__()
        if unsupported or find_unused_parameters or process_group is not None:
            raise NotImplementedError("avid_hip.parallel.DistributedDataParallel: unsupported argument(s) "
                                      f"{sorted(unsupported) or ['find_unused_parameters / process_group']}")
        if device_ids is not None and len(device_ids) != 1:
            raise ValueError("one process per GPU: device_ids must name one device")
        self.module = module
        self.device_ids = device_ids
        self.broadcast_buffers = bool(broadcast_buffers)
        cap = int((bucket_cap_mb if bucket_cap_mb is not None else 16) * (1 << 20))
        self._engine = _WrapperEngine(module, cap, self.broadcast_buffers)

    def forward(self, *inputs, **kwargs):
        if not (self.training and torch.is_grad_enabled()):
            return self.module(*inputs, **kwargs)
        from . import plan
        eng = self._engine
        eng.begin_step()
        if eng.broadcast_buffers == "step":
            eng.sync_buffers()
        with plan.engine(eng):
            return self.module(*inputs, **kwargs)

    def no_sync(self):
        raise NotImplementedError("avid_hip.parallel.DistributedDataParallel: no_sync() (gradient accumulation) is not reproduced")


class Adam(torch.optim.Optimizer):
    """`torch.optim.Adam(params, lr, betas, eps, weight_decay)` (utils/main_utils.py:250-256; L2 weight decay, no amsgrad) whose
    `step()` is ONE launch over the flat parameter / gradient / moment buffers.  A `torch.optim.Optimizer`: `param_groups`
    (MultiStepLR writes `lr` there, utils/main_utils.py:258), `state_dict()` / `load_state_dict()` in torch.optim.Adam's own
    format (`state[i] = {step, exp_avg, exp_avg_sq}`), so checkpoints written by either load into the other.

    The parameters are used where they lie if they already are the views of one flat buffer (the model went through
    `DistributedDataParallel` above or `TrainStep`), otherwise they are re-seated into one here.  One parameter group.

    Differences from torch.optim.Adam, by construction of the one-launch step: `zero_grad()` always zero-fills (set_to_none is
    accepted and ignored: `.grad` stays the view of the flat buffer), and `step()` updates EVERY parameter of the buffer — one
    that received no gradient this step is stepped with a zero gradient (its moments decay, L2 weight decay still applies),
    where torch.optim.Adam would skip a parameter whose `.grad` is None.  The reference freezes nothing (main-avid.py:106)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False, **unsupported):
        if amsgrad or any(unsupported.get(k) for k in ("maximize", "capturable", "differentiable")):
            raise NotImplementedError("avid_hip.parallel.Adam: amsgrad / maximize / capturable / differentiable")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay, amsgrad=False))
        if len(self.param_groups) != 1:
            raise NotImplementedError("avid_hip.parallel.Adam: one parameter group")
        ps = self.param_groups[0]["params"]
        self.flat = flat_of(ps)
        if self.flat is None:
            # Some of them may already BE views of a live flat buffer (the model went through DistributedDataParallel / TrainStep)
            # while the list is not exactly that buffer's set (extra trainable criterion parameters, a subset): re-seating them
            # into a second buffer here would orphan the first — the launch programs and collectives would keep writing the
            # engine's gradient buffer while this optimizer read its own zeros: a silent no-learning failure.
            owned = [p for p in ps if _FLAT_OF.get(id(p)) is not None and _FLAT_OF[id(p)]() is not None]
            if owned:
                raise ValueError("avid_hip.parallel.Adam: %d of the %d parameters already live in a flat buffer (DistributedDataParallel / "
                                 "TrainStep) but the list is not exactly that buffer's parameter set (or they were moved after it was built); "
                                 "pass exactly the wrapped model's parameters, or use torch.optim.Adam for a different set" % (len(owned), len(ps)))
            self.flat = FlatParams(ps)
        self.m = torch.zeros_like(self.flat.flat)
        self.v = torch.zeros_like(self.flat.flat)
        self._t = 0
        self._step_t = torch.tensor(0.0)

    def _views(self, i):
        p, o = self.flat.params[i], self.flat.offsets[i]
        return (self.m[o:o + p.numel()].as_strided(p.shape, p.stride()), self.v[o:o + p.numel()].as_strided(p.shape, p.stride()))

    def _publish_state(self):
        """`self.state` in torch.optim.Adam's shape: views of the flat moments, one shared step tensor."""
        if self.state:
            return
        for i, p in enumerate(self.flat.params):
            m, v = self._views(i)
            self.state[p] = {"step": self._step_t, "exp_avg": m, "exp_avg_sq": v}

    def zero_grad(self, set_to_none=True):
        """One fill of the flat gradient buffer; `.grad` stays seated (a launch program writes the buffer, not the attribute)."""
        self.flat.grad.zero_()
        self.flat.seat_grads()

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        from . import ops
        g = self.param_groups[0]
        p0 = self.flat.params[0]
        if p0.data_ptr() != self.flat.flat.data_ptr() + 4 * self.flat.offsets[0]:
            raise RuntimeError("avid_hip.parallel.Adam: the parameters were moved after the optimizer was built (model.cuda() / a "
                               "wrapper constructed later): build the optimizer last, as main-avid.py:95-108 does")
        self.flat.seat_grads()
        self._t += 1
        self._step_t.fill_(float(self._t))
        ops.adam_flat(self.flat.flat, self.flat.grad, self.m, self.v, g["lr"], g["betas"][0], g["betas"][1], g["eps"],
                      g["weight_decay"], self._t)
        self._publish_state()
        return loss

    def state_dict(self):
        if self._t > 0:
            self._publish_state()
        return super().state_dict()

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)              # (torch: casts, re-keys by parameter, replaces self.state)
        self.m.zero_()
        self.v.zero_()
        step = 0
        for i, p in enumerate(self.flat.params):
            st = self.state.get(p)
            if not st:
                continue
            m, v = self._views(i)
            m.copy_(st["exp_avg"])
            v.copy_(st["exp_avg_sq"])
            step = max(step, int(float(st["step"])))
        self._t = step
        self._step_t = torch.tensor(float(step))
        self.state.clear()
        if step > 0:
            self._publish_state()
