"""torch.autograd.Function wrappers around the C-ABI (include/avid_hip.h).

PyTorch is plumbing here: it owns device memory, the current HIP stream and the autograd graph.
Every op below launches hand-written gfx950 kernels through ctypes; none has a torch/CPU fallback —
a CPU tensor raises ``AvidHipError``.

Layout contract: activations are contiguous channels-last ``[B, T, H, W, C]``;
conv / linear weights keep the reference's logical shape ``[Cout, Cin, kt, kh, kw]`` but live in
memory as ``[Cout][kt][kh][kw][Cin]`` (torch ``channels_last_3d``); see ``make_weight``.
"""
from __future__ import annotations

import ctypes as C
import os
import struct

import torch
from torch.autograd import Function

from . import lib
from .lib import AvidHipError, ConvDesc

_WS = {}
OVERLAP_IN_CAPTURE = True      # the two towers stay on two streams inside a hipGraph capture (graph branches)
# Weight gradients on a trailing stream (steps driven by parallel.TrainStep / a launch program): nothing in the backward
# chain consumes dw, so wgrad(L) is issued on a helper stream that only waits for its dy and is joined once, before
# the optimizer — it then runs next to the small kernels of the chain (BatchNorm finalize / apply, split-K reduces)
# instead of in front of them.  Needs the gradient to land in the flat buffer (GradSlots): a dw handed back to
# autograd would be accumulated on the main stream before the helper stream has written it.  With or without a process
# group (round 3 kept it to single-process steps: what it measured then was two hardware queues on one dispatch pipe,
# avid_hip/streams.py).
DEFER_WGRAD = int(__import__('os').environ.get('AVID_DEFER_WGRAD', '1'))
# (inside a hipGraph capture the trailing streams become graph branches; the replay schedules them worse than the
# eager streams: 4518 -> 4309 clips/s with --graph 1, so a capture keeps the weight gradients on the compute streams)
DEFER_IN_CAPTURE = False
_DEFER_ON = False          # set by parallel.TrainStep around loss.backward()
_DEFERRED = {}             # compute stream handle -> its trailing wgrad stream
_DEFER_USED = set()


def wgrad_stream(device):
    """(current compute stream, the trailing weight-gradient stream of its StreamSet) — ONE trailing stream serves both
    towers: four streams in all, one per dispatch pipe (avid_hip/streams.py)."""
    from . import streams
    ss = streams.current_set(device)
    _DEFERRED[(device.index, ss.trail.cuda_stream)] = ss.trail
    return torch.cuda.current_stream(device), ss.trail


def join_deferred_wgrads():
    """The current stream waits for every trailing weight-gradient stream used since the last join."""
    if _DEFER_USED:
        cur = torch.cuda.current_stream()
        for st in _DEFER_USED:
            cur.wait_stream(st)
        _DEFER_USED.clear()


# NOTE: a deferred / queued weight gradient READS dy and x after the autograd node that produced dy has
# returned, on a trailing stream or in a later grouped launch.  The same dy tensor may be handed on to autograd (as the
# gradient of a fused addend, or as the dy of the BatchNorm upstream).  That is safe because every consumer of it in the
# shipped models only reads it; a user graph in which autograd ACCUMULATES IN PLACE into that tensor (a second consumer
# of the addend whose gradient arrives first and is then added to) could race with the trailing read: record_stream only
# guards the allocation.  Deferral and grouping are therefore confined to steps driven by parallel.TrainStep (GradSlots
# armed), whose models have single-consumer gradients; elsewhere weight gradients run in line.
# ------------------------------------------------------------------------------------------------
# grouped weight gradients: the small layers' wgrads of a backward pass are queued (per compute stream) and launched a
# dozen at a time as ONE persistent kernel (avid_conv_wgrad_group) — only where the gradients land in the flat buffer
# (GradSlots armed by parallel.TrainStep), because the launch happens later than the autograd node that produced dy
# ------------------------------------------------------------------------------------------------
GROUP_WGRAD = int(os.environ.get("AVID_GROUP_WGRAD", "1"))
GROUP_MAX = 12           # items of one grouped launch (avid_conv_wgrad_group takes at most 12; round-3 sweep: flat within noise)
# queued layers that are flushed when a layer with its own weight-gradient launch comes by
GROUP_MIN_FLUSH = 4
_GROUP_PENDING = {}        # compute stream handle -> [(desc, x, dy, dst, slot)]
_GROUP_WS_BYTES = {}       # tuple of layer geometries -> workspace bytes


def _group_key(device):
    return (device.index, _raw_stream(device.index) if _raw_stream is not None
            else torch.cuda.current_stream(device).cuda_stream)


def queue_wgrad(d, x, dy, dst, slot):
    key = _group_key(x.device)
    lst = _GROUP_PENDING.setdefault(key, [])
    lst.append((d, x, dy, dst, slot))
    if len(lst) >= GROUP_MAX:
        flush_wgrad_group(x.device)


def flush_wgrad_group(device=None, all_streams=False):
    """Launch the queued weight gradients of the current stream (``all_streams``: of every stream, each on its own —
    the caller has made the current stream wait for them or will) and report their parameters as ready."""
    keys = list(_GROUP_PENDING) if all_streams else [_group_key(device)]
    for key in keys:
        lst = _GROUP_PENDING.get(key)
        if not lst:
            continue
        _GROUP_PENDING[key] = []
        _launch_wgrad_group(lst, key)


def _launch_wgrad_group(lst, key):
    n = len(lst)
    items = (lib.WgradItem * n)()
    for i, (d, x, dy, dst, _) in enumerate(lst):
        items[i].d = d
        items[i].x, items[i].dy, items[i].dw = x.data_ptr(), dy.data_ptr(), dst.data_ptr()
    geo = tuple(bytes(t[0]) for t in lst)           # (the layers' geometries)
    nb = _GROUP_WS_BYTES.get(geo)
    if nb is None:
        nb = _GROUP_WS_BYTES[geo] = lib.raw("avid_conv_wgrad_group_workspace_bytes")(n, items)
    dev = lst[0][1].device
    cur_handle = _group_key(dev)[1]
    if cur_handle == key[1]:
        stream_ctx = None
    else:                                           # flushing another compute stream's queue: launch it over there
        stream_ctx = torch.cuda.stream(torch.cuda.ExternalStream(key[1], device=dev))

    def launch():
        capturing = torch.cuda.is_current_stream_capturing()
        if (_DEFER_ON and (DEFER_IN_CAPTURE or not capturing)):
            main, trail = wgrad_stream(dev)
            trail.wait_stream(main)                 # every dy of the group is complete on its compute stream
            with torch.cuda.stream(trail):
                ws = workspace(dev, nb)
                lib.call("avid_conv_wgrad_group", n, items, _p(ws), ws.numel(), _stream())
                for t in lst:                       # (inside the context: the bucket's producer is the trailing stream)
                    _grad_done(t[4])
            if not capturing:
                for _, x, dy, _, _ in lst:
                    x.record_stream(trail)
                    dy.record_stream(trail)
            _DEFER_USED.add(trail)
        else:
            ws = workspace(dev, nb)
            lib.call("avid_conv_wgrad_group", n, items, _p(ws), ws.numel(), _stream())
            for t in lst:
                _grad_done(t[4])
    if stream_ctx is None:
        launch()
    else:
        ext = torch.cuda.ExternalStream(key[1], device=dev)
        with stream_ctx:
            launch()
        torch.cuda.current_stream(dev).wait_stream(ext)       # (a late flush of the other tower's queue)


class deferred_wgrads:
    """``with ops.deferred_wgrads(): loss.backward()`` — weight gradients may trail on helper streams inside; they
    are joined on exit."""

    def __init__(self, enabled=True):
        self.enabled = enabled

    def __enter__(self):
        global _DEFER_ON
        # (timed launches stay on one stream)
        self.prev, _DEFER_ON = _DEFER_ON, bool(DEFER_WGRAD) and self.enabled and not lib.TIMING
        return self

    def __exit__(self, *exc):
        global _DEFER_ON
        flush_wgrad_group(all_streams=True)        # (while the trailing streams are still on)
        _DEFER_ON = self.prev
        join_deferred_wgrads()
        return False


def side_stream(device, slot=1):
    """The audio tower's stream of the current compute stream (placed on its own dispatch pipe: avid_hip/streams.py)."""
    from . import streams
    return streams.current_set(device).side


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)   # ~0.3 us; current_stream() builds a Stream object (~8 us)


def _stream():
    if _raw_stream is not None:
        return C.c_void_p(_raw_stream(torch.cuda.current_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise AvidHipError("avid_hip ops need HIP device tensors — there is no CPU fallback")


def workspace(device, nbytes):
    """Stream-ordered scratch, grown on demand, one per (device, stream)."""
    key = (device.index, _raw_stream(device.index) if _raw_stream is not None
           else torch.cuda.current_stream(device).cuda_stream)
    ws = _WS.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _WS[key] = ws
    return ws


# ------------------------------------------------------------------------------------------------
# device-side index errors (include/avid_hip.h AVID_DEVERR_*)
# ------------------------------------------------------------------------------------------------
class DeviceErrors:
    """One int32 error word per device that the gather / scatter kernels OR a code into when they meet an id
    outside [0, N) — where the reference's indexing raises (criterions/avid.py:57-62,124; avid_cma.py:199).

    ``poll()`` never synchronises: it looks at the copy of the word that the PREVIOUS poll sent to pinned host
    memory (if that copy has landed), raises ``IndexError`` if it is non-zero, and starts the next copy.  The
    criterion polls at the start and at the end of every forward, so a corrupt ``index`` surfaces within about one
    step (as soon as the host sees the copy that followed the offending kernels); ``check()`` is the blocking
    form for callers that synchronise anyway (``loss.item()``)."""
    _MSG = {1: "bank_scores: negative / positive index outside [0, num_data)",
            2: "update_memory: sample index outside [0, num_data)",
            4: "memory_sampling: sample index outside [0, num_data) (positive_set lookup)"}
    _inst = {}

    def __init__(self, device):
        self.flag = torch.zeros((), dtype=torch.int32, device=device)
        self.host = torch.zeros((), dtype=torch.int32).pin_memory()
        self.event = None

    @classmethod
    def get(cls, device):
        inst = cls._inst.get(device.index)
        if inst is None:
            inst = cls._inst[device.index] = cls(device)
        return inst

    def ptr(self):
        return C.c_void_p(self.flag.data_ptr())

    def _raise(self, code):
        self.flag.zero_()
        self.host.zero_()
        self.event = None
        what = "; ".join(m for b, m in self._MSG.items() if code & b) or f"code {code}"
        raise IndexError(f"avid_hip: index out of range in a device kernel — {what}")

    def poll(self):
        if torch.cuda.is_current_stream_capturing():
            return
        if self.event is not None:
            if not self.event.query():
                return                      # the previous copy is still in flight: look again next time
            code = int(self.host)
            if code:
                self._raise(code)
        self.host.copy_(self.flag, non_blocking=True)
        self.event = torch.cuda.Event()
        self.event.record()

    def check(self):
        code = int(self.flag.item())
        if code:
            self._raise(code)


def poll_device_errors(device):
    DeviceErrors.get(device).poll()


def check_device_errors(device=None):
    """Blocking check of the device error word (synchronises the current stream)."""
    device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
    DeviceErrors.get(device).check()


# ------------------------------------------------------------------------------------------------
# weights
# ------------------------------------------------------------------------------------------------
def make_weight(cout, cin, *k):
    """Uninitialised tensor of logical shape [cout, cin, *k] whose memory is [cout, *k, cin]."""
    return torch.empty(cout, *k, cin).movedim(-1, 1)


def weight_layout_ok(w):
    return w.dim() == 2 and w.is_contiguous() or (w.dim() > 2 and w.movedim(1, -1).is_contiguous())


def _kdims(w):
    k = tuple(w.shape[2:])
    return (1,) * (3 - len(k)) + k


_DESC_CACHE = {}
_BN_WS_CACHE = {}


def _desc_cached(xs, cin, cout, k, stride, pad, channel_first):
    """(ConvDesc, fwd_ws_bytes, dgrad_ws_bytes, wgrad_ws_bytes, bn_partial_rows) — once per distinct layer geometry.
    (``d.bn_bwd_rows``: rows of BatchNorm-backward partials its dgrad can write, 0 = cannot.)"""
    key = (xs, cin, cout, k, stride, pad, channel_first,
           torch.cuda.current_device() if torch.cuda.is_available() else -1)       # (plans depend on the CU count)
    hit = _DESC_CACHE.get(key)
    if hit is None:
        d = _desc(xs, cin, cout, k, stride, pad, channel_first)
        hit = (d, lib.raw("avid_conv_fwd_workspace_bytes")(C.byref(d)),
               0 if channel_first else lib.raw("avid_conv_dgrad_workspace_bytes")(C.byref(d)),
               lib.raw("avid_conv_wgrad_workspace_bytes")(C.byref(d)),
               lib.raw("avid_conv_fwd_stats_rows")(C.byref(d)))
        d.bn_bwd_rows = 0 if channel_first else lib.raw("avid_conv_dgrad_bn_rows")(C.byref(d))
        d.groupable = bool(lib.raw("avid_conv_wgrad_groupable")(C.byref(d)))
        d.wino_fwd = int(lib.raw("avid_conv_uses_wino")(C.byref(d), 0))        # 0, 1 (wino_kernel) or 2 (wino2_kernel)
        d.wino_dgrad = int(lib.raw("avid_conv_uses_wino")(C.byref(d), 1))
        # the 128 x 64 implicit-GEMM tile takes its weights pre-split into bf16 terms (avid_wt_desc mode 5 / 6)
        d.split_fwd = bool(lib.raw("avid_conv_uses_split")(C.byref(d), 0))
        d.split_dgrad = False if channel_first else bool(lib.raw("avid_conv_uses_split")(C.byref(d), 1))
        # the layer can read the INPUT of the BatchNorm (+ReLU) in front of it and apply the map while staging (forward and weight
        # gradient): avid_conv_fwd_in / avid_conv_wgrad_in — the launch programs then never write the normalised tensor
        d.in_affine = bool(lib.raw("avid_conv_takes_in_affine")(C.byref(d)))
        _DESC_CACHE[key] = hit
    return hit


_WINO_EPOCH = [0]


def wino_epoch():
    """Bumped whenever the Winograd dispatch switches change: compiled launch programs (avid_hip.plan) are keyed by it."""
    return _WINO_EPOCH[0]


def wino_configure(enabled=-1, min_pixels=-1, max_channels=-1):
    """Dispatch switches of the Winograd kernels (``avid_wino_configure``; negative = environment / default) — and
    drop the per-layer plans cached here, which depend on them."""
    lib.call("avid_wino_configure", int(enabled), int(min_pixels), int(max_channels))
    _DESC_CACHE.clear()
    _GROUP_WS_BYTES.clear()
    _WINO_EPOCH[0] += 1


def wino2_configure(min_rounds_x10=-1):
    """Which Winograd forward / input-gradient kernel a layer takes (``avid_wino2_configure``): 0 = always
    ``wino2_kernel``, negative = environment / default (layers with >= 1.5 rounds of 64-tile units)."""
    lib.call("avid_wino2_configure", int(min_rounds_x10))
    _DESC_CACHE.clear()
    _GROUP_WS_BYTES.clear()
    _WINO_EPOCH[0] += 1


def set_cu_budget(cus=0):
    """CUs the persistent kernels plan for (``avid_set_cu_budget``; 0 = all): leaves the rest to co-running kernels such as
    RCCL's collectives.  Drops the cached per-layer plans and bumps the epoch the compiled launch programs are keyed by;
    returns the effective count."""
    got = int(lib.raw("avid_set_cu_budget")(int(cus)))
    _DESC_CACHE.clear()
    _GROUP_WS_BYTES.clear()
    _WINO_EPOCH[0] += 1
    return got


def cu_budget():
    return int(lib.raw("avid_cu_budget")())


def _bn_ws_bytes(M, Cc):
    key = (M, Cc)
    nb = _BN_WS_CACHE.get(key)
    if nb is None:
        nb = _BN_WS_CACHE[key] = lib.raw("avid_bn_workspace_bytes")(M, Cc)
    return nb


def _desc(xs, cin, cout, k, stride, pad, channel_first):
    B, Ti, Hi, Wi = xs
    d = ConvDesc()
    d.B, d.Ti, d.Hi, d.Wi, d.Cin = B, Ti, Hi, Wi, cin
    d.kt, d.kh, d.kw = k
    d.st, d.sh, d.sw = stride
    d.pt, d.ph, d.pw = pad
    d.To = (Ti + 2 * pad[0] - k[0]) // stride[0] + 1
    d.Ho = (Hi + 2 * pad[1] - k[1]) // stride[1] + 1
    d.Wo = (Wi + 2 * pad[2] - k[2]) // stride[2] + 1
    d.Cout = cout
    d.x_channel_first = 1 if channel_first else 0
    return d


class GradSlots:
    """Parameter gradients written straight into their slice of a flat gradient buffer.

    While ``armed()``, a backward kernel that produces a parameter gradient gets the parameter's slice of the
    flat buffer as its output (``dst``) and the autograd.Function returns ``None`` for that input — no
    AccumulateGrad add kernel per parameter (141 tiny launches per step otherwise).  A parameter that is hit a
    second time in the same backward (shared weights) falls back to a fresh tensor and normal accumulation
    into ``p.grad`` (which is the same slice).  ``on_ready(i)`` is called for parameter i when its gradient
    has been issued (gradient-bucket all-reduce)."""

    def __init__(self, params, grad_views, on_ready=None):
        self.index = {p.data_ptr(): i for i, p in enumerate(params)}
        self.views = grad_views
        self.on_ready = on_ready
        self.used = set()

    def has_slot(self, param_ptr):
        """True if the gradient of the parameter at ``param_ptr`` would be written into the flat buffer now."""
        i = self.index.get(param_ptr)
        return i is not None and i not in self.used

    def armed(self):
        return _ArmSlots(self)


class _ArmSlots:
    def __init__(self, gs):
        self.gs = gs

    def __enter__(self):
        global _SLOTS
        self.prev, _SLOTS = _SLOTS, self.gs
        self.gs.used.clear()
        return self.gs

    def __exit__(self, *exc):
        global _SLOTS
        # weight gradients still queued for a grouped launch belong to THIS arming (ops.deferred_wgrads flushes them
        # itself; a caller that arms the slots alone gets them here): launch them, or drop them after an exception —
        # never leave (x, dy) of this step to a later step's flush
        if exc and exc[0] is not None:
            _GROUP_PENDING.clear()
        elif any(_GROUP_PENDING.values()):
            flush_wgrad_group(all_streams=True)
        _SLOTS = self.prev
        return False


_SLOTS = None


def _grad_dst(param_ptr, like=None, shape=None, device=None):
    """(tensor to write the gradient of the parameter at ``param_ptr`` into, slot index or None)."""
    gs = _SLOTS
    if gs is not None:
        i = gs.index.get(param_ptr)
        if i is not None and i not in gs.used:
            gs.used.add(i)
            return gs.views[i], i
    if like is not None:
        return torch.empty_like(like), None
    return torch.empty(shape, dtype=torch.float32, device=device), None


def _grad_done(i):
    if i is not None and _SLOTS is not None and _SLOTS.on_ready is not None:
        _SLOTS.on_ready(i)


U_FLOATS = 24        # a Winograd-transformed weight table: 96 bytes per (Cout, Cin) pair (avid_wt_desc modes 1 - 4)


class TransposedWeights:
    """[Cin][taps][Cout] copies of every conv / linear weight that can need an input gradient, repacked by ONE
    launch (avid_weight_transpose_batched) instead of one small launch inside every avid_conv_dgrad call.

    The copies are only trusted while ``armed()`` is active: ``TrainStep`` refreshes them right before
    ``loss.backward()`` and arms them for its duration, so weights changed in any way between steps
    (optimizer, load_state_dict, manual edits) are always picked up."""

    def __init__(self, params):
        ws = [p for p in params
              if p.is_cuda and p.dtype == torch.float32 and p.dim() in (2, 4, 5) and weight_layout_ok(p)
              and p.shape[1] % 64 == 0 and p.shape[0] % 32 == 0]
        self.n = len(ws)
        self.map = {}
        if not ws:
            return
        dev = ws[0].device
        self.buf = torch.empty(sum(p.numel() for p in ws), dtype=torch.float32, device=dev)
        recs, off, self.max_elems = [], 0, 0
        for p in ws:
            k = _kdims(p)
            wt = self.buf[off:off + p.numel()]
            off += p.numel()
            recs.append(struct.pack("<QQiiii", p.data_ptr(), wt.data_ptr(), p.shape[0], k[0] * k[1] * k[2], p.shape[1], 0))
            self.map[p.data_ptr()] = wt
            self.max_elems = max(self.max_elems, p.numel())
        self.table = torch.frombuffer(bytearray(b"".join(recs)), dtype=torch.uint8).to(dev)
        # Winograd-transformed weights U (wino.hip): which 3x3 layers run on that path, in which direction and on which
        # of the two kernels (their operand-fragment orders differ) depends on the activations' size, i.e. is known at
        # call time: the first step's calls REGISTER what they need (`_u_for`) and transform inside the call; from the
        # second step on one launch per step keeps the INPUT GRADIENTS' transforms current (`refresh_wino`, on the helper
        # stream) and the calls pick them up; the forward's stay inside the calls: hoisted in front of the model they delay
        # the stem convolution, which then loses its race with the helper stream, and behind the gradient fill on the
        # helper the first Winograd layer waits for them (round 3: forward 4.25 -> 4.37 ms, no net gain; 410f826, db70d52).
        self.wino_on = True
        self.wino_fwd_on = False
        self.umap, self.uwant, self.n_wino, self.ucount = {}, {}, 0, [0, 0]
        self.uready, self.uevent, self.uwaited = False, None, set()
        self.wparams = {p.data_ptr(): p for p in ws if _kdims(p) == (1, 3, 3)}

    def want_wino(self, w, code):
        """Called by a convolution that found no current transform of ``w`` (code 1 / 2: forward / input gradient in
        wino_kernel's fragment order, 3 / 4: in wino2_kernel's): have one from the next ``refresh_wino`` on."""
        if self.wino_on and w.data_ptr() in self.wparams:
            self.uwant[(w.data_ptr(), code)] = True

    def _build_wino(self):
        keys = sorted(set(self.umap) | set(self.uwant))
        self.uwant = {}
        ps = [self.wparams[k[0]] for k in keys]
        dev = ps[0].device
        self.ubuf = torch.empty(sum(U_FLOATS * p.shape[0] * p.shape[1] for p in ps), dtype=torch.float32, device=dev)
        recs, off, self.umap = {0: [], 1: []}, 0, {}
        for (ptr, code), p in zip(keys, ps):
            u = self.ubuf[off:off + U_FLOATS * p.shape[0] * p.shape[1]]
            off += u.numel()
            recs[1 - (code & 1)].append(struct.pack("<QQiiii", p.data_ptr(), u.data_ptr(), p.shape[0], 9, p.shape[1], code))
            self.umap[(ptr, code)] = u
        # two descriptor tables: [0] the forward's transforms (codes 1, 3), [1] the input gradient's (2, 4)
        self.utable = [torch.frombuffer(bytearray(b"".join(r)), dtype=torch.uint8).to(dev) if r else None for r in (recs[0], recs[1])]
        self.ucount = [len(recs[0]), len(recs[1])]
        self.umax = max(p.shape[0] * p.shape[1] for p in ps)
        self.n_wino = len(keys)

    def refresh(self):
        if self.n:
            lib.call("avid_weight_transpose_batched", self.n, _p(self.table), self.max_elems, _stream())

    def refresh_wino(self, backward):
        """The Winograd transforms of the current weights for the forward (backward = False) or the input gradients
        (True): one launch on the current stream."""
        if not self.wino_on:
            return
        if self.n_wino and self.ucount[int(backward)]:
            lib.call("avid_weight_transpose_batched", self.ucount[int(backward)], _p(self.utable[int(backward)]), self.umax,
                     _stream())
        if not backward:
            # consumers on other streams wait for this launch at their first lookup of the step (_u_for)
            self.uready = True
            self.uevent = torch.cuda.Event()
            self.uevent.record()
            self.uwaited = {torch.cuda.current_stream().cuda_stream}

    def prepare_wino(self):
        """Rebuild the table if the previous step registered new layers — before EITHER refresh of this step, or one of
        them fills the old buffers while the calls read the new ones."""
        if self.uwant and getattr(self, "frozen", False):
            # a captured graph holds the addresses of the current table (its transform launches write it, its
            # convolutions read it): layers that register later transform their weights inside their own calls
            self.uwant = {}
        elif self.uwant and not torch.cuda.is_current_stream_capturing():
            self._build_wino()

    def armed_wino(self):
        return _ArmWino(self)

    def armed(self):
        return _ArmTransposed(self)


class _ArmTransposed:
    def __init__(self, tw):
        self.tw = tw

    def __enter__(self):
        global _TRANSPOSED
        self.prev, _TRANSPOSED = _TRANSPOSED, self.tw
        return self.tw

    def __exit__(self, *exc):
        global _TRANSPOSED
        _TRANSPOSED = self.prev
        return False


class _ArmWino:
    def __init__(self, tw):
        self.tw = tw

    def __enter__(self):
        global _WINO_U
        self.prev, _WINO_U = _WINO_U, self.tw if self.tw.wino_on else None
        return self.tw

    def __exit__(self, *exc):
        global _WINO_U
        _WINO_U = self.prev
        return False


_TRANSPOSED = None
_WINO_U = None

def _u_for(w, mode, variant):
    """This weight's Winograd transform (mode 1 forward, 2 input gradient; variant 1 wino_kernel, 2 wino2_kernel) if a
    TrainStep keeps one current; otherwise None — the call transforms the weights itself — and, inside a TrainStep, a
    request to have it from the next step on."""
    tw = _WINO_U
    if tw is None:
        return None
    if mode == 1 and not tw.wino_fwd_on:
        return None
    code = mode + 2 * (variant - 1)
    u = tw.umap.get((w.data_ptr(), code))
    if u is None:
        tw.want_wino(w, code)
        return None
    if mode == 1:                        # forward: refreshed on the helper stream behind the stem convolution
        if not tw.uready:
            return None
        cur = torch.cuda.current_stream()
        if cur.cuda_stream not in tw.uwaited:
            cur.wait_event(tw.uevent)
            tw.uwaited.add(cur.cuda_stream)
    return u


_SPLIT_CACHE = {}


def _split_for(w, mode):
    """``w`` pre-split into three bf16 terms in igemm_pk_kernel's fragment order (avid_wt_desc mode 5: for the forward,
    6: the transpose, for the input gradient): made by one small launch per call on this per-layer path (the launch
    programs of avid_hip.plan keep every layer's in their per-step table).  Same terms either way: bit-identical results."""
    key = (w.data_ptr(), tuple(w.shape), mode)
    planes = _SPLIT_CACHE.get(key)
    if planes is None:
        planes = torch.empty(6 * w.numel(), dtype=torch.uint8, device=w.device)
        if len(_SPLIT_CACHE) >= 1024:
            _SPLIT_CACHE.clear()
        _SPLIT_CACHE[key] = planes
    k = _kdims(w)
    desc = lib.WtDesc(w.data_ptr(), planes.data_ptr(), w.shape[0], k[0] * k[1] * k[2], w.shape[1], mode)
    lib.call("avid_weight_transform", C.addressof(desc), _stream())
    return planes


def _fwd_u(w, d, plain=True):
    """`u` of avid_conv_fwd; ``plain``: no bias / ReLU in the epilogue (those variants read the fp32 weights)."""
    if d.wino_fwd:
        return _u_for(w, 1, d.wino_fwd)
    return _split_for(w, 5) if d.split_fwd and plain else None


def _dgrad_u(w, d):
    if d.wino_dgrad:
        return _u_for(w, 2, d.wino_dgrad)
    return _split_for(w, 6) if d.split_dgrad else None


def _wt_for(w):
    if _TRANSPOSED is None:
        return None
    return _TRANSPOSED.map.get(w.data_ptr())


FUSE_BN_BWD = os.environ.get("AVID_FUSE_BN_BWD", "1") == "1"


class BnSource:
    """Hand-over between a training-mode BatchNorm(+ReLU) and the convolution that is the SOLE consumer of its
    output: the conv's input-gradient kernel then also produces the BatchNorm's backward partial sums (while the
    gradient is in registers) and leaves them here; the BatchNorm's backward, which autograd runs next, takes
    them instead of making its own pass over dy and x.  One object per forward call."""
    __slots__ = ("x", "stats4", "relu", "partials", "dx_ptr", "dx_ver")

    def __init__(self, x, stats4, relu):
        self.x, self.stats4, self.relu, self.partials, self.dx_ptr, self.dx_ver = x, stats4, relu, None, 0, 0


class _ConvCL(Function):
    """y = conv(x, w) [+ addend] [+ bias] [relu]  — avid_conv_fwd / avid_conv_dgrad / avid_conv_wgrad."""

    @staticmethod
    def forward(ctx, x, w, addend, bias, stride, pad, relu, channel_first, want_stats=False, tap=False, bn_src=None,
                res_w=None, res_stride=None):
        _need_cuda(x, w, addend, bias, res_w)
        ctx.bn_src = bn_src
        if not x.is_contiguous():
            raise AvidHipError("conv: x must be contiguous (channels-last [B,T,H,W,C])")
        if not weight_layout_ok(w):
            raise AvidHipError("conv: weight memory is not [Cout][k...][Cin] (channels-last); "
                               "was the parameter re-created with .contiguous()?")
        if x.dtype != torch.float32 or w.dtype != torch.float32:
            raise AvidHipError("conv: fp32 only")
        k = _kdims(w)
        cout, cin = w.shape[0], w.shape[1]
        if channel_first:
            B, c, Ti, Hi, Wi = x.shape
        else:
            B, Ti, Hi, Wi, c = x.shape
        if c != cin:
            raise AvidHipError(f"conv: input has {c} channels, weight expects {cin}")
        d, nb, ctx.nb_dgrad, ctx.nb_wgrad, srows = _desc_cached((B, Ti, Hi, Wi), cin, cout, k, stride, pad, channel_first)
        y = torch.empty((B, d.To, d.Ho, d.Wo, cout), dtype=torch.float32, device=x.device)
        if addend is not None and (addend.shape != y.shape or not addend.is_contiguous()):
            raise AvidHipError("conv: addend must be a contiguous tensor of the output shape")
        ws = workspace(x.device, nb) if nb else None
        # BatchNorm partial sums of y from the conv epilogue ([rows][2][Cout]; None: this layer cannot)
        stats = None
        if want_stats and srows > 0 and bias is None and not relu:
            stats = torch.empty((srows, 2, cout), dtype=torch.float32, device=x.device)
        lib.call("avid_conv_fwd", C.byref(d), _p(x), _p(w), _p(_fwd_u(w, d, bias is None and not relu)), _p(addend), _p(bias),
                 int(relu), _p(y), _p(stats), _p(ws),
                 ws.numel() if ws is not None else 0, _stream())
        ctx.d, ctx.relu, ctx.channel_first = d, relu, channel_first
        ctx.has_addend, ctx.has_bias = addend is not None, bias is not None
        ctx.bias_ptr = bias.data_ptr() if bias is not None else 0
        # The block's 1x1x1 strided residual convolution on the same input (models/network_blocks.py:47-51): computed
        # here so that its input gradient can stay COMPACT (the sub-sampled grid it reads) and ride in this op's
        # dgrad as a sparse addend, instead of being scattered into an x-shaped tensor of mostly zeros first.
        ctx.has_res = res_w is not None
        y_res = None
        if ctx.has_res:
            if tap or channel_first or _kdims(res_w) != (1, 1, 1) or res_w.shape[1] != cin or not weight_layout_ok(res_w):
                raise AvidHipError("conv: the fused residual convolution must be 1x1x1 over the same channels-last input")
            rs = tuple(int(v) for v in res_stride)
            dr, nbr, _, ctx.nb_wgrad_r, _ = _desc_cached((B, Ti, Hi, Wi), cin, res_w.shape[0], (1, 1, 1), rs, (0, 0, 0), False)
            y_res = torch.empty((B, dr.To, dr.Ho, dr.Wo, res_w.shape[0]), dtype=torch.float32, device=x.device)
            wsr = workspace(x.device, nbr) if nbr else None
            lib.call("avid_conv_fwd", C.byref(dr), _p(x), _p(res_w), _p(_fwd_u(res_w, dr)), None, None, 0, _p(y_res), None, _p(wsr),
                     wsr.numel() if wsr is not None else 0, _stream())
            ctx.dr, ctx.res_stride = dr, rs
        ctx.save_for_backward(x, w, y if relu else None, res_w)
        ctx.want_stats, ctx.tap = want_stats, tap
        ctx.set_materialize_grads(False)   # no zero-fill kernel for the outputs that get no gradient
        outs = [y]
        if want_stats:
            if stats is None:
                stats = torch.empty(0, dtype=torch.float32, device=x.device)
            ctx.mark_non_differentiable(stats)
            outs.append(stats)
        if tap:
            # an alias of the input for a second consumer (the residual branch): its gradient comes back into
            # THIS backward and rides in the dgrad kernel's addend — no separate accumulate kernel
            outs.append(x.view(x.shape))
        if ctx.has_res:
            outs.append(y_res)
        return outs[0] if len(outs) == 1 else tuple(outs)

    @staticmethod
    def backward(ctx, dy, *more):
        x, w, y, res_w = ctx.saved_tensors
        d = ctx.d
        d_tap = more[-1] if (ctx.tap and more) else None
        d_res = more[-1] if (ctx.has_res and more) else None
        if dy is None and d_res is None:   # only the tap carried a gradient
            return d_tap, None, None, None, None, None, None, None, None, None, None, None, None
        if dy is None:
            raise AvidHipError("conv: the main output of a convolution with a fused residual branch got no gradient")
        dy = dy.contiguous()
        if d_tap is not None:
            d_tap = d_tap.contiguous()
        st = _stream()
        if ctx.relu:
            g = torch.empty_like(dy)
            lib.call("avid_relu_bwd", dy.numel(), _p(y), _p(dy), _p(g), st)
            dy = g
        dx = dw = dadd = dbias = None
        need_dx, need_dw = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        if need_dx and ctx.channel_first:
            raise AvidHipError("conv: input gradient of a channel-first stem is not implemented (never needed)")

        def run_wgrad():
            ws = workspace(x.device, ctx.nb_wgrad)
            g, slot = _grad_dst(w.data_ptr(), like=w)   # the flat-buffer slice, or preserve_format empty_like
            if g.stride() != w.stride() and not weight_layout_ok(g):
                raise AvidHipError("conv: the weight-gradient tensor does not have the [Cout][k][Cin] layout")
            lib.call("avid_conv_wgrad", C.byref(d), _p(x), _p(dy), _p(g), _p(ws), ws.numel(), _stream())
            if slot is not None:
                _grad_done(slot)
                return None
            return g

        deferred = False
        capturing = torch.cuda.is_current_stream_capturing()
        grouped = False
        if need_dw and GROUP_WGRAD and _SLOTS is not None and d.groupable and _SLOTS.has_slot(w.data_ptr()):
            # a small layer: its weight gradient rides in the next grouped launch of this compute stream
            g, slot = _grad_dst(w.data_ptr(), like=w)
            if g.stride() != w.stride() and not weight_layout_ok(g):
                raise AvidHipError("conv: the weight-gradient tensor does not have the [Cout][k][Cin] layout")
            queue_wgrad(d, x, dy, g, slot)
            dw = None
            need_dw = False
            grouped = True
        elif need_dw and GROUP_WGRAD and len(_GROUP_PENDING.get(_group_key(x.device), ())) >= GROUP_MIN_FLUSH:
            # a large layer (own launch): the small layers queued so far go first — their gradient buckets should not wait
            # for the end of the backward pass
            flush_wgrad_group(x.device)
        if grouped:
            pass
        elif (need_dw and _DEFER_ON and _SLOTS is not None and (DEFER_IN_CAPTURE or not capturing)
                and _SLOTS.has_slot(w.data_ptr())):
            main, trail = wgrad_stream(x.device)
            trail.wait_stream(main)            # dy (after the ReLU mask) is complete
            with torch.cuda.stream(trail):
                dw = run_wgrad()
            if not capturing:                  # (a capture's private pool frees nothing before the graph is done)
                x.record_stream(trail)
                dy.record_stream(trail)
            _DEFER_USED.add(trail)
            deferred = True
        add, add_stride, dw_res = d_tap, None, None
        if d_res is not None:
            d_res = d_res.contiguous()
            dr = ctx.dr
            if need_dx:
                # compact input gradient of the residual convolution: a dense 1x1x1 dgrad over the sub-sampled grid
                dc, _, nbc, _, _ = _desc_cached((d.B, dr.To, dr.Ho, dr.Wo), d.Cin, dr.Cout, (1, 1, 1), (1, 1, 1), (0, 0, 0),
                                                False)
                add = torch.empty((d.B, dr.To, dr.Ho, dr.Wo, d.Cin), dtype=torch.float32, device=x.device)
                wsc = workspace(x.device, nbc)
                lib.call("avid_conv_dgrad", C.byref(dc), _p(d_res), _p(res_w), _p(_wt_for(res_w)), _p(_dgrad_u(res_w, dc)), None, None, _p(add), None,
                         _p(wsc), wsc.numel(), st)
                if any(v != 1 for v in ctx.res_stride):
                    add_stride = (C.c_int32 * 3)(*ctx.res_stride)
            if ctx.needs_input_grad[11]:
                def run_wgrad_res():
                    wsr = workspace(x.device, ctx.nb_wgrad_r)
                    g, slot = _grad_dst(res_w.data_ptr(), like=res_w)
                    lib.call("avid_conv_wgrad", C.byref(dr), _p(x), _p(d_res), _p(g), _p(wsr), wsr.numel(), _stream())
                    if slot is not None:
                        _grad_done(slot)
                        return None
                    return g
                if GROUP_WGRAD and _SLOTS is not None and dr.groupable and _SLOTS.has_slot(res_w.data_ptr()):
                    g, slot = _grad_dst(res_w.data_ptr(), like=res_w)
                    queue_wgrad(dr, x, d_res, g, slot)
                elif deferred and _SLOTS.has_slot(res_w.data_ptr()):
                    main, trail = wgrad_stream(x.device)
                    trail.wait_stream(main)
                    with torch.cuda.stream(trail):
                        dw_res = run_wgrad_res()
                    if not capturing:
                        d_res.record_stream(trail)
                else:
                    dw_res = run_wgrad_res()
        if need_dx:
            ws = workspace(x.device, ctx.nb_dgrad)
            dx = torch.empty_like(x)
            fuse = None
            src = ctx.bn_src
            if src is not None and FUSE_BN_BWD and d.bn_bwd_rows > 0 and src.x.shape == x.shape:
                # dx is the whole gradient of the BatchNorm output x: its backward partial sums ride along
                src.partials = torch.empty((d.bn_bwd_rows, 2, d.Cin), dtype=torch.float32, device=x.device)
                # the BatchNorm's backward verifies that THIS tensor, unmodified, is its dy
                src.dx_ptr, src.dx_ver = dx.data_ptr(), dx._version
                s4 = src.stats4
                fuse = lib.BnBwdFuse(_p(src.x), _p(s4[2]), _p(s4[3]), _p(s4[0]), _p(s4[1]), int(src.relu),
                                     _p(src.partials))
            lib.call("avid_conv_dgrad", C.byref(d), _p(dy), _p(w), _p(_wt_for(w)),
                     _p(_dgrad_u(w, d)), _p(add),
                     add_stride, _p(dx),
                     C.byref(fuse) if fuse is not None else None, _p(ws), ws.numel(), st)
        if need_dw and not deferred and not grouped:
            dw = run_wgrad()
        if ctx.has_addend and ctx.needs_input_grad[2]:
            dadd = dy
        if ctx.has_bias and ctx.needs_input_grad[3]:
            dbias, slot = _grad_dst(ctx.bias_ptr, shape=(d.Cout,), device=dy.device)
            lib.call("avid_colsum", dy.numel() // d.Cout, d.Cout, _p(dy), _p(dbias), st)
            if slot is not None:
                _grad_done(slot)
                dbias = None
        if d_tap is not None and not need_dx:
            dx = d_tap
        return dx, dw, dadd, dbias, None, None, None, None, None, None, None, dw_res, None


def conv_fwd_in(x, w, stride, pad, scale, shift, relu=True, addend=None, bn_stats=False):
    """``avid_conv_fwd_in`` as a plain call (no autograd): conv(ReLU?(x * scale + shift), w) [+ addend] with x the INPUT of the
    BatchNorm in front of the layer — what the launch programs emit for conv2x's temporal layers.  Returns y, or (y, partials).
    Raises for a layer that cannot (``_desc_cached(...)[0].in_affine``)."""
    _need_cuda(x, w, addend, scale, shift)
    B, Ti, Hi, Wi, cin = x.shape
    d, nb, _, _, srows = _desc_cached((B, Ti, Hi, Wi), cin, w.shape[0], _kdims(w), tuple(stride), tuple(pad), False)
    y = torch.empty((B, d.To, d.Ho, d.Wo, w.shape[0]), dtype=torch.float32, device=x.device)
    stats = torch.empty((srows, 2, w.shape[0]), dtype=torch.float32, device=x.device) if bn_stats and srows > 0 else None
    ws = workspace(x.device, nb) if nb else None
    aff = lib.InAffine(scale.data_ptr(), shift.data_ptr(), int(bool(relu)))
    lib.call("avid_conv_fwd_in", C.byref(d), _p(x), C.byref(aff), _p(w), _p(_fwd_u(w, d)), _p(addend), None, 0, _p(y), _p(stats),
             _p(ws), ws.numel() if ws is not None else 0, _stream())
    return (y, stats) if bn_stats else y


def conv_wgrad_in(x, dy, w_like, stride, pad, scale, shift, relu=True):
    """``avid_conv_wgrad_in`` as a plain call: the weight gradient of conv(ReLU?(x * scale + shift), w) given dy."""
    _need_cuda(x, dy, scale, shift)
    B, Ti, Hi, Wi, cin = x.shape
    d, _, _, nbw, _ = _desc_cached((B, Ti, Hi, Wi), cin, w_like.shape[0], _kdims(w_like), tuple(stride), tuple(pad), False)
    dw = torch.empty_like(w_like)
    ws = workspace(x.device, nbw) if nbw else None
    aff = lib.InAffine(scale.data_ptr(), shift.data_ptr(), int(bool(relu)))
    lib.call("avid_conv_wgrad_in", C.byref(d), _p(x), C.byref(aff), _p(dy), _p(dw), _p(ws), ws.numel() if ws is not None else 0,
             _stream())
    return dw


def conv_cl(x, w, stride=(1, 1, 1), pad=(0, 0, 0), addend=None, bias=None, relu=False, channel_first=False,
            bn_stats=False, tap=False, bn_src=None, res=None):
    """``bn_stats=True`` adds ``partials`` to the result: y's BatchNorm partial sums from the conv epilogue (pass
    them to ``batch_norm_cl``), or an empty tensor when the layer cannot produce them.  ``tap=True`` adds an
    alias of ``x`` for a second consumer whose gradient is then summed inside this op's dgrad kernel.
    ``bn_src`` (a ``BnSource``): x is the output of that BatchNorm and this conv (with its tap, if any) is the
    only consumer — the BatchNorm's backward partial sums are then produced by this op's dgrad kernel.
    ``res = (res_weight, res_stride)``: the block's 1x1x1 strided residual convolution of the same x is computed
    too (last element of the result); its input gradient stays compact and is added inside this op's dgrad."""
    if res is not None:      # ``res = (weight, stride)``: also return the 1x1x1 strided residual convolution of x
        return _ConvCL.apply(x, w, addend, bias, tuple(stride), tuple(pad), bool(relu), bool(channel_first),
                             bool(bn_stats), bool(tap), bn_src, res[0], tuple(res[1]))
    if bn_stats or tap or bn_src is not None:
        return _ConvCL.apply(x, w, addend, bias, tuple(stride), tuple(pad), bool(relu), bool(channel_first),
                             bool(bn_stats), bool(tap), bn_src)
    return _ConvCL.apply(x, w, addend, bias, tuple(stride), tuple(pad), bool(relu), bool(channel_first))


def linear(x, w, bias=None, relu=False):
    """nn.Linear (+ReLU) as a 1x1x1 conv over [B,1,1,1,C] — models/av_wrapper.py:23-29."""
    B, Cin = x.shape
    y = conv_cl(x.reshape(B, 1, 1, 1, Cin), w, (1, 1, 1), (0, 0, 0), None, bias, relu, False)
    return y.reshape(B, w.shape[0])


# ------------------------------------------------------------------------------------------------
# BatchNorm (+ReLU)
# ------------------------------------------------------------------------------------------------
class _BatchNormCL(Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, stats, training, momentum, eps, relu, counter=None, partials=None, src=None):
        _need_cuda(x, gamma, beta)
        ctx.src = src
        if not x.is_contiguous():
            raise AvidHipError("bn: x must be contiguous channels-last")
        Cc = x.shape[-1]
        M = x.numel() // Cc
        rm, rv = stats
        y = torch.empty_like(x)
        st = _stream()
        if training:
            stats4 = torch.empty((4, Cc), dtype=torch.float32, device=x.device)   # mean, invstd, scale, shift
            ws = workspace(x.device, _bn_ws_bytes(M, Cc))
            lib.call("avid_bn_fwd_train", M, Cc, _p(x), _p(gamma), _p(beta), _p(rm), _p(rv), float(momentum),
                     float(eps), int(relu), _p(y), _p(stats4[0]), _p(stats4[1]), _p(stats4[2]), _p(stats4[3]),
                     _p(counter), _p(partials), 0 if partials is None else partials.shape[0], _p(ws), ws.numel(), st)
            ctx.save_for_backward(x, gamma, stats4)
            ctx.beta_ptr = beta.data_ptr()
            if src is not None:
                src.x, src.stats4, src.relu = x, stats4, relu
        else:
            # eval mode; when a gradient can flow (fine-tuning with frozen BatchNorm) the coefficients are kept
            grad = any(ctx.needs_input_grad[:3])
            stats4 = torch.empty((4, Cc), dtype=torch.float32, device=x.device) if grad else None
            lib.call("avid_bn_fwd_eval", M, Cc, _p(x), _p(gamma), _p(beta), _p(rm), _p(rv), float(eps), int(relu),
                     _p(y), _p(stats4), st)
            if grad:
                ctx.save_for_backward(x, gamma, stats4)
                ctx.beta_ptr = beta.data_ptr()
            else:
                ctx.save_for_backward()
        ctx.training, ctx.relu, ctx.M, ctx.C = training, relu, M, Cc
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, stats4 = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        dgamma, sg = _grad_dst(gamma.data_ptr(), shape=(ctx.C,), device=x.device)
        dbeta, sb = _grad_dst(ctx.beta_ptr, shape=(ctx.C,), device=x.device)
        ws = workspace(x.device, _bn_ws_bytes(ctx.M, ctx.C))
        part = None
        if ctx.src is not None:           # left by the dgrad kernel that produced dy (see BnSource)
            part, ctx.src.partials = ctx.src.partials, None
            # The hand-over is only valid if that dgrad's output IS this BatchNorm's whole output gradient.  A
            # second consumer of the activation (a hook, a custom block, an intermediate in the loss) makes
            # autograd sum the gradients — into a new tensor, or IN PLACE into the dgrad's output (which bumps its
            # version counter): then the partial sums cover one branch only and the unfused pass over dy and x
            # runs instead.
            if part is not None and (dy.data_ptr() != ctx.src.dx_ptr or dy._version != ctx.src.dx_ver):
                part = None
        lib.call("avid_bn_bwd", ctx.M, ctx.C, _p(x), _p(dy), _p(gamma), _p(stats4[0]), _p(stats4[1]), _p(stats4[2]),
                 _p(stats4[3]), int(ctx.relu), _p(dx), _p(dgamma), _p(dbeta), _p(part),
                 0 if part is None else part.shape[0], 0 if ctx.training else 1, _p(ws), ws.numel(), _stream())
        if sg is not None:
            _grad_done(sg)
            dgamma = None
        if sb is not None:
            _grad_done(sb)
            dbeta = None
        return dx, dgamma, dbeta, None, None, None, None, None, None, None, None


def batch_norm_cl(x, gamma, beta, running_mean, running_var, training, momentum=0.1, eps=1e-5, relu=False,
                  num_batches_tracked=None, partials=None, src=None):
    """``num_batches_tracked`` (0-d int64 device tensor or None) is bumped by the kernel in training mode;
    ``partials``: x's partial sums from ``conv_cl(..., bn_stats=True)`` (skips the statistics pass)."""
    if partials is not None and (partials.numel() == 0 or not training):
        partials = None
    return _BatchNormCL.apply(x, gamma, beta, (running_mean, running_var), bool(training), momentum, eps, bool(relu),
                              num_batches_tracked, partials, src if training else None)


class _BnReluMaxPool(Function):
    """maxpool_hw3s2(relu(bn_train(x))) in one pass over x (the video stem's tail, models/video.py:21-23)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, stats, momentum, eps, counter, partials=None):
        _need_cuda(x, gamma, beta)
        if not x.is_contiguous():
            raise AvidHipError("bn_relu_maxpool: x must be contiguous channels-last")
        nparts = 0 if partials is None or partials.numel() == 0 else partials.shape[0]
        B, T, H, W, Cc = x.shape
        Ho, Wo = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
        rm, rv = stats
        y = torch.empty((B, T, Ho, Wo, Cc), dtype=torch.float32, device=x.device)
        am = torch.empty((B, T, Ho, Wo, Cc), dtype=torch.uint8, device=x.device)
        stats4 = torch.empty((4, Cc), dtype=torch.float32, device=x.device)   # mean, invstd, scale, shift
        ws = workspace(x.device, _bn_ws_bytes(B * T * H * W, Cc))
        lib.call("avid_bn_relu_maxpool_fwd", B, T, H, W, Cc, _p(x), _p(gamma), _p(beta), _p(rm), _p(rv),
                 float(momentum), float(eps), _p(y), _p(am), _p(stats4[0]), _p(stats4[1]), _p(stats4[2]), _p(stats4[3]),
                 _p(counter), _p(partials) if nparts else None, nparts, _p(ws), ws.numel(), _stream())
        ctx.save_for_backward(x, gamma, stats4, am)
        ctx.beta_ptr = beta.data_ptr()
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, stats4, am = ctx.saved_tensors
        B, T, H, W, Cc = x.shape
        dx = torch.empty_like(x)
        dgamma, sg = _grad_dst(gamma.data_ptr(), shape=(Cc,), device=x.device)
        dbeta, sb = _grad_dst(ctx.beta_ptr, shape=(Cc,), device=x.device)
        ws = workspace(x.device, _bn_ws_bytes(B * T * H * W, Cc))
        lib.call("avid_bn_relu_maxpool_bwd", B, T, H, W, Cc, _p(x), _p(dy.contiguous()), _p(am), _p(gamma),
                 _p(stats4[0]), _p(stats4[1]), _p(stats4[2]), _p(stats4[3]), _p(dx), _p(dgamma), _p(dbeta), _p(ws),
                 ws.numel(), _stream())
        if sg is not None:
            _grad_done(sg)
            dgamma = None
        if sb is not None:
            _grad_done(sb)
            dbeta = None
        return dx, dgamma, dbeta, None, None, None, None, None


def bn_relu_maxpool(x, gamma, beta, running_mean, running_var, momentum=0.1, eps=1e-5, num_batches_tracked=None,
                    partials=None):
    """Training-mode BatchNorm + ReLU + MaxPool (1,3,3)/(1,2,2)/(0,1,1) fused (the normalised activation is
    never written; backward rebuilds the un-pooled gradient from the argmax slots).  ``partials``: the BatchNorm
    partial sums the stem convolution produced with its output (conv_cl(..., bn_stats=True))."""
    return _BnReluMaxPool.apply(x, gamma, beta, (running_mean, running_var), momentum, eps, num_batches_tracked,
                                partials)


# ------------------------------------------------------------------------------------------------
# video front end: ClipToTensor + Normalize on the GPU
# ------------------------------------------------------------------------------------------------
def clip_normalize(frames, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225)):
    """``frames [B, T, H, W, 3]`` uint8 on the GPU -> ``[B, 3, T, H, W]`` fp32 = ((u / 255) - mean) / std — the
    reference's ``ClipToTensor`` + ``Normalize`` (datasets/preprocessing.py:45-48), bit-identical.  No gradient."""
    _need_cuda(frames)
    if frames.dim() != 5 or frames.shape[-1] != 3 or frames.dtype != torch.uint8 or not frames.is_contiguous():
        raise AvidHipError("clip_normalize: frames must be a contiguous uint8 [B, T, H, W, 3] tensor")
    B, T_, H, W, _ = frames.shape
    out = torch.empty((B, 3, T_, H, W), dtype=torch.float32, device=frames.device)
    m = (C.c_float * 3)(*[float(v) for v in mean])
    s = (C.c_float * 3)(*[float(v) for v in std])
    lib.call("avid_clip_normalize", B, T_, H, W, _p(frames), m, s, _p(out), _stream())
    return out


# ------------------------------------------------------------------------------------------------
# audio front end
# ------------------------------------------------------------------------------------------------
_LOGSPEC_BASIS = {}


def log_spectrogram(sig, n_stft, hop, frames, mean=None, std=None, top_db=100.0):
    """``sig [B, L]`` mono fp32 on the GPU -> ``[B, 1, frames, n_stft/4 + 1]`` (avid_logspec; reference
    datasets/preprocessing.py:174-186 with librosa's stft / power_to_db defaults).  No gradient."""
    _need_cuda(sig, mean, std)
    if sig.dim() != 2 or sig.dtype != torch.float32 or not sig.is_contiguous():
        raise AvidHipError("log_spectrogram: sig must be a contiguous fp32 [B, L] tensor")
    B, L = sig.shape
    key = (sig.device, int(n_stft))
    basis = _LOGSPEC_BASIS.get(key)
    if basis is None:
        nfl = lib.raw("avid_logspec_basis_floats")(int(n_stft))
        if nfl == 0:
            raise AvidHipError(f"log_spectrogram: unsupported STFT size {n_stft}")
        basis = torch.empty(nfl, dtype=torch.float32, device=sig.device)
        lib.call("avid_logspec_basis", int(n_stft), _p(basis), _stream())
        _LOGSPEC_BASIS[key] = basis
    out = torch.empty((B, 1, int(frames), n_stft // 4 + 1), dtype=torch.float32, device=sig.device)
    ws = workspace(sig.device, lib.raw("avid_logspec_workspace_bytes")(B, int(n_stft), int(frames)))
    lib.call("avid_logspec", B, L, _p(sig), int(n_stft), int(hop), int(frames), _p(basis), _p(mean), _p(std),
             float(top_db), _p(out), _p(ws), ws.numel(), _stream())
    return out


# ------------------------------------------------------------------------------------------------
# pooling
# ------------------------------------------------------------------------------------------------
class _MaxPoolHW3S2(Function):
    @staticmethod
    def forward(ctx, x):
        _need_cuda(x)
        B, T, H, W, Cc = x.shape
        Ho, Wo = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
        y = torch.empty((B, T, Ho, Wo, Cc), dtype=torch.float32, device=x.device)
        am = torch.empty((B, T, Ho, Wo, Cc), dtype=torch.uint8, device=x.device)
        lib.call("avid_maxpool_hw3s2_fwd", B, T, H, W, Cc, _p(x.contiguous()), _p(y), _p(am), _stream())
        ctx.save_for_backward(am)
        ctx.shape = (B, T, H, W, Cc)
        return y

    @staticmethod
    def backward(ctx, dy):
        (am,) = ctx.saved_tensors
        B, T, H, W, Cc = ctx.shape
        dx = torch.empty(ctx.shape, dtype=torch.float32, device=dy.device)
        lib.call("avid_maxpool_hw3s2_bwd", B, T, H, W, Cc, _p(dy.contiguous()), _p(am), _p(dx), _stream())
        return dx


def maxpool_hw3s2(x):
    return _MaxPoolHW3S2.apply(x)


class _GlobalMaxPool(Function):
    @staticmethod
    def forward(ctx, x):
        _need_cuda(x)
        x = x.contiguous()
        B, Cc = x.shape[0], x.shape[-1]
        S = x.numel() // (B * Cc)
        y = torch.empty((B, Cc), dtype=torch.float32, device=x.device)
        am = torch.empty((B, Cc), dtype=torch.int32, device=x.device)
        lib.call("avid_global_maxpool_fwd", B, S, Cc, _p(x), _p(y), _p(am), _stream())
        ctx.save_for_backward(am)
        ctx.shape = tuple(x.shape)
        ctx.S = S
        return y

    @staticmethod
    def backward(ctx, dy):
        (am,) = ctx.saved_tensors
        B, Cc = ctx.shape[0], ctx.shape[-1]
        dx = torch.empty(ctx.shape, dtype=torch.float32, device=dy.device)
        lib.call("avid_global_maxpool_bwd", B, ctx.S, Cc, _p(dy.contiguous()), _p(am), _p(dx), _stream())
        return dx


def global_maxpool(x):
    """AdaptiveMaxPool{2,3}d(1) over a channels-last tensor -> [B, C]."""
    return _GlobalMaxPool.apply(x)


# ------------------------------------------------------------------------------------------------
# criterion ops
# ------------------------------------------------------------------------------------------------
class _L2Norm(Function):
    @staticmethod
    def forward(ctx, x):
        _need_cuda(x)
        x = x.contiguous()
        bs, D = x.shape
        y = torch.empty_like(x)
        nrm = torch.empty(bs, dtype=torch.float32, device=x.device)
        lib.call("avid_l2norm_fwd", bs, D, _p(x), _p(y), _p(nrm), _stream())
        ctx.save_for_backward(y, nrm)
        return y

    @staticmethod
    def backward(ctx, dy):
        y, nrm = ctx.saved_tensors
        bs, D = y.shape
        dx = torch.empty_like(y)
        lib.call("avid_l2norm_bwd", bs, D, _p(y), _p(nrm), _p(dy.contiguous()), _p(dx), _stream())
        return dx


def l2_normalize(x):
    return _L2Norm.apply(x)


class _BankScores(Function):
    """scores[b][j] = <bank[idx[b][j]], emb[b]> / T ; the bank gets no gradient (avid.py:56)."""

    @staticmethod
    def forward(ctx, emb, bank, idx, inv_T):
        _need_cuda(emb, bank, idx)
        need_rows = ctx.needs_input_grad[0]      # (decided before .contiguous(): its copy would not require grad)
        emb = emb.contiguous()
        idx = idx.contiguous()
        if idx.dtype != torch.int64 or not bank.is_contiguous():
            raise AvidHipError("bank_scores: idx must be int64 and the bank contiguous")
        bs, D = emb.shape
        R = idx.shape[1]
        s = torch.empty((bs, R), dtype=torch.float32, device=emb.device)
        # The reference's autograd keeps the PRE-update rows (the bank is EMA-updated inside forward,
        # criterions/avid.py:78, before backward runs): snapshot them while they stream through.
        rows = torch.empty((bs, R, D), dtype=torch.float32, device=emb.device) if need_rows else None
        lib.call("avid_bank_scores_fwd", bs, R, D, bank.shape[0], _p(idx), _p(bank), _p(emb), float(inv_T), _p(s),
                 _p(rows), DeviceErrors.get(emb.device).ptr(), _stream())
        ctx.save_for_backward(rows)
        ctx.inv_T, ctx.dims = inv_T, (bs, R, D)
        return s

    @staticmethod
    def backward(ctx, ds):
        (rows,) = ctx.saved_tensors
        bs, R, D = ctx.dims
        demb = torch.empty((bs, D), dtype=torch.float32, device=ds.device)
        lib.call("avid_bank_scores_bwd", bs, R, D, 0, _p(rows), None, None, _p(ds.contiguous()),
                 float(ctx.inv_T), 0, _p(demb), _stream())
        return demb, None, None, None


def bank_scores(emb, bank, idx, inv_T):
    return _BankScores.apply(emb, bank, idx, inv_T)


def mean_exp(s):
    """mean(exp(s)) of a (possibly column-sliced) 2-D score matrix -> 0-d device tensor."""
    _need_cuda(s)
    rows, cols = s.shape
    if s.stride(1) != 1:
        s = s.contiguous()
    out = torch.empty((), dtype=torch.float32, device=s.device)
    lib.call("avid_mean_exp", rows, cols, s.stride(0), _p(s), _p(out), _stream())
    return out


_NCE_WS = {}


def _nce_ws(device):
    """Per-(device, stream) scratch of avid_nce_fwd: zero-filled once, then the kernel re-arms its own ticket.
    Keyed by stream as well: two streams must not share the partial sums of a launch in flight."""
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    ws = _NCE_WS.get(key)
    if ws is None:
        ws = torch.zeros(int(lib.raw("avid_nce_workspace_bytes")()), dtype=torch.uint8, device=device)
        _NCE_WS[key] = ws
    return ws


class _NCELoss(Function):
    @staticmethod
    def forward(ctx, spos, sneg, Z):
        _need_cuda(spos, sneg, Z)
        if spos.stride(1) != 1:
            spos = spos.contiguous()
        if sneg.stride(1) != 1:
            sneg = sneg.contiguous()
        bs, P = spos.shape
        K = sneg.shape[1]
        loss = torch.empty((), dtype=torch.float32, device=spos.device)
        ws = _nce_ws(spos.device)
        lib.call("avid_nce_fwd", bs, P, K, _p(spos), spos.stride(0), _p(sneg), sneg.stride(0), _p(Z), 1.0, 0,
                 _p(loss), _p(ws), ws.numel(), _stream())
        ctx.save_for_backward(spos, sneg, Z)
        return loss

    @staticmethod
    def backward(ctx, dloss):
        spos, sneg, Z = ctx.saved_tensors
        bs, P = spos.shape
        K = sneg.shape[1]
        dpos = torch.empty((bs, P), dtype=torch.float32, device=spos.device)
        dneg = torch.empty((bs, K), dtype=torch.float32, device=spos.device)
        lib.call("avid_nce_bwd", bs, P, K, _p(spos), spos.stride(0), _p(sneg), sneg.stride(0), _p(Z),
                 _p(dloss.contiguous()), 1.0, _p(dpos), _p(dneg), _stream())
        return dpos, dneg, None


class _NCELossJoint(Function):
    """NCE over one score tensor ``s [bs, P + K]`` whose first P columns are the positives (bank_scores' layout).
    Same kernels as _NCELoss; the gradient goes back as ONE tensor — through ``s[:, :P]`` / ``s[:, P:]`` autograd
    builds it from two zero-filled tensors, two slice copies and an add (5 small kernels per NCE term on the
    chain between forward and backward)."""

    @staticmethod
    def forward(ctx, s, P, Z):
        _need_cuda(s, Z)
        if s.stride(1) != 1:
            s = s.contiguous()
        bs, K = s.shape[0], s.shape[1] - P
        spos, sneg = s[:, :P], s[:, P:]
        loss = torch.empty((), dtype=torch.float32, device=s.device)
        ws = _nce_ws(s.device)
        lib.call("avid_nce_fwd", bs, P, K, _p(spos), spos.stride(0), _p(sneg), sneg.stride(0), _p(Z), 1.0, 0,
                 _p(loss), _p(ws), ws.numel(), _stream())
        ctx.save_for_backward(s, Z)
        ctx.P = P
        return loss

    @staticmethod
    def backward(ctx, dloss):
        s, Z = ctx.saved_tensors
        P = ctx.P
        bs, K = s.shape[0], s.shape[1] - P
        spos, sneg = s[:, :P], s[:, P:]
        dpos = torch.empty((bs, P), dtype=torch.float32, device=s.device)
        dneg = torch.empty((bs, K), dtype=torch.float32, device=s.device)
        lib.call("avid_nce_bwd", bs, P, K, _p(spos), spos.stride(0), _p(sneg), sneg.stride(0), _p(Z),
                 _p(dloss.contiguous()), 1.0, _p(dpos), _p(dneg), _stream())
        return torch.cat([dpos, dneg], 1), None, None


def nce_loss(spos, sneg, Z):
    # the two views of one bank_scores result (criterions/avid.py tags them): differentiate the joint tensor
    joint = getattr(spos, "_avid_joint", None)
    if (joint is not None and getattr(sneg, "_avid_joint", None) is joint and joint.is_cuda
            and spos.shape[1] + sneg.shape[1] == joint.shape[1] and spos.data_ptr() == joint.data_ptr()):
        return _NCELossJoint.apply(joint, spos.shape[1], Z)
    return _NCELoss.apply(spos, sneg, Z)


def split_scores(s, P):
    """``[s[:, :P], s[:, P:]]`` (the reference's [positives, negatives] pair, criterions/avid.py:70-75) tagged with
    the tensor they are views of, so that ``nce_loss`` can take the joint path."""
    pos, neg = s[:, :P], s[:, P:]
    pos._avid_joint = s
    neg._avid_joint = s
    return [pos, neg]


def alias_draw(n, K, prob, alias, uniform, seed, offset, y=None, per_row=1, device=None, offset_dev=None):
    """``offset_dev`` (optional 0-d int64 device tensor): the draw counter is read from it and advanced
    on the stream — required for hipGraph replay, where by-value arguments are frozen."""
    device = device if device is not None else (y.device if y is not None else prob.device)
    if device.type != "cuda":
        raise AvidHipError("alias_draw: HIP device required")
    out = torch.empty(n, dtype=torch.int64, device=device)
    st = _stream()
    lib.call("avid_alias_draw", n, K, _p(prob), _p(alias), int(uniform), int(seed), int(offset), _p(offset_dev),
             _p(y), int(per_row), _p(out), st)
    if offset_dev is not None:
        lib.call("avid_counter_add", _p(offset_dev), 1, st)
    return out


def bank_update(bank, y, emb, momentum):
    _need_cuda(bank, y, emb)
    lib.call("avid_bank_update", y.shape[0], bank.shape[1], bank.shape[0], _p(bank), _p(y.contiguous()),
             _p(emb.contiguous()), float(momentum), DeviceErrors.get(bank.device).ptr(), _stream())


def bank_update_pair(bank0, bank1, y, emb0, emb1, momentum0, momentum1):
    """``bank_update`` of both banks in one launch (criterions/avid.py:118-129), bit-identical to two calls."""
    _need_cuda(bank0, bank1, y, emb0, emb1)
    lib.call("avid_bank_update2", y.shape[0], bank0.shape[1], bank0.shape[0], _p(bank0), _p(bank1), _p(y.contiguous()),
             _p(emb0.contiguous()), _p(emb1.contiguous()), float(momentum0), float(momentum1),
             DeviceErrors.get(bank0.device).ptr(), _stream())


FUSED_CRITERION = os.environ.get("AVID_FUSED_CRITERION", "1") == "1"


class _XModalFused(Function):
    """The whole cross-modal criterion of a steady-state step in one kernel (``avid_xmodal_fused``): returns
    (total loss, losses[4], v_hat, a_hat); the gradient of the total loss with respect to both raw embeddings is formed
    in the forward pass (Z is a constant) and only scaled by the upstream gradient in backward."""

    @staticmethod
    def forward(ctx, v_emb, a_emb, y, idx, bank_v, bank_a, Z, inv_T, coeff, ws):
        _need_cuda(v_emb, a_emb, y, idx, bank_v, bank_a, Z, ws)
        v_emb, a_emb, y, idx = v_emb.contiguous(), a_emb.contiguous(), y.contiguous(), idx.contiguous()
        bs, D = v_emb.shape
        K = idx.shape[1]
        if y.dtype != torch.int64 or idx.dtype != torch.int64 or not bank_v.is_contiguous() or not bank_a.is_contiguous():
            raise AvidHipError("xmodal_fused: y / idx must be int64 and the banks contiguous")
        dev = v_emb.device
        hats = torch.empty((2, bs, D), dtype=torch.float32, device=dev)
        grads = torch.empty((2, bs, D), dtype=torch.float32, device=dev)
        losses = torch.empty(4, dtype=torch.float32, device=dev)
        lib.call("avid_xmodal_fused", bs, K, D, bank_v.shape[0], _p(v_emb), _p(a_emb), _p(y), _p(idx), _p(bank_v),
                 _p(bank_a), float(inv_T), _p(Z), float(coeff), _p(hats[0]), _p(hats[1]), _p(losses), _p(grads[0]),
                 _p(grads[1]), _p(ws), ws.numel(), DeviceErrors.get(dev).ptr(), _stream())
        ctx.save_for_backward(grads)
        total = losses[3]
        ctx.mark_non_differentiable(losses, hats)
        return total, losses, hats

    @staticmethod
    def backward(ctx, dtotal, _dl, _dh):
        (grads,) = ctx.saved_tensors
        g = grads * dtotal                      # one launch for both embeddings
        return g[0], g[1], None, None, None, None, None, None, None, None


def xmodal_fused_workspace(device, bs, K):
    """Zero-filled scratch of ``avid_xmodal_fused`` (its device tickets re-arm themselves): one per criterion object."""
    return torch.zeros(int(lib.raw("avid_xmodal_fused_workspace_bytes")(int(bs), int(K))), dtype=torch.uint8, device=device)


def xmodal_fused(v_emb, a_emb, y, idx, bank_v, bank_a, Z, inv_T, coeff, ws):
    return _XModalFused.apply(v_emb, a_emb, y, idx, bank_v, bank_a, Z, inv_T, coeff, ws)


class _CMAFused(Function):
    """The AVID+CMA criterion of a steady-state step (cross-modal instance + within-modal positive terms) in one kernel
    (``avid_cma_fused``): returns (total loss, losses[8], hats); as ``_XModalFused`` the gradient with respect to both raw
    embeddings is formed in the forward pass and only scaled in backward."""

    @staticmethod
    def forward(ctx, v_emb, a_emb, y, pos, idx, bank_v, bank_a, Z, inv_T, Kw, coeff_inst, coeff_pos, ws):
        _need_cuda(v_emb, a_emb, y, pos, idx, bank_v, bank_a, Z, ws)
        v_emb, a_emb, y, pos, idx = v_emb.contiguous(), a_emb.contiguous(), y.contiguous(), pos.contiguous(), idx.contiguous()
        bs, D = v_emb.shape
        P, K = pos.shape[1], idx.shape[1]
        if any(t.dtype != torch.int64 for t in (y, pos, idx)) or not bank_v.is_contiguous() or not bank_a.is_contiguous():
            raise AvidHipError("cma_fused: y / pos / idx must be int64 and the banks contiguous")
        dev = v_emb.device
        hats = torch.empty((2, bs, D), dtype=torch.float32, device=dev)
        grads = torch.empty((2, bs, D), dtype=torch.float32, device=dev)
        losses = torch.empty(8, dtype=torch.float32, device=dev)
        lib.call("avid_cma_fused", bs, P, K, int(Kw), D, bank_v.shape[0], _p(v_emb), _p(a_emb), _p(y), _p(pos), _p(idx),
                 _p(bank_v), _p(bank_a), float(inv_T), _p(Z), float(coeff_inst), float(coeff_pos), _p(hats[0]), _p(hats[1]),
                 _p(losses), _p(grads[0]), _p(grads[1]), _p(ws), ws.numel(), DeviceErrors.get(dev).ptr(), _stream())
        ctx.save_for_backward(grads)
        total = losses[6]
        ctx.mark_non_differentiable(losses, hats)
        return total, losses, hats

    @staticmethod
    def backward(ctx, dtotal, _dl, _dh):
        (grads,) = ctx.saved_tensors
        g = grads * dtotal
        return (g[0], g[1]) + (None,) * 11


def cma_fused_workspace(device, bs, P, K):
    return torch.zeros(int(lib.raw("avid_cma_fused_workspace_bytes")(int(bs), int(P), int(K))), dtype=torch.uint8, device=device)


def cma_fused(v_emb, a_emb, y, pos, idx, bank_v, bank_a, Z, inv_T, Kw, coeff_inst, coeff_pos, ws):
    return _CMAFused.apply(v_emb, a_emb, y, pos, idx, bank_v, bank_a, Z, inv_T, Kw, coeff_inst, coeff_pos, ws)


def cma_negatives(positive_set, y, rand_idx):
    _need_cuda(positive_set, y, rand_idx)
    bs, K = rand_idx.shape
    P = positive_set.shape[1]
    pos = torch.empty((bs, P), dtype=torch.int64, device=y.device)
    neg = torch.empty((bs, K), dtype=torch.int64, device=y.device)
    lib.call("avid_cma_negatives", bs, K, P, positive_set.shape[0], _p(positive_set), _p(y.contiguous()),
             _p(rand_idx.contiguous()), _p(pos), _p(neg), DeviceErrors.get(y.device).ptr(), _stream())
    return pos, neg


def adam_flat(p, g, m, v, lr, beta1, beta2, eps, wd, step, grad_scale=1.0, step_dev=None, lr_dev=None, advance=True):
    """``step_dev`` (optional 0-d int64 device tensor): advanced by one on the stream, then read by the
    kernel for the bias corrections (hipGraph-replay safe); otherwise ``step`` is used by value.
    ``lr_dev`` (optional 0-d fp32 device tensor): the learning rate is read from it (a captured graph freezes
    ``lr``; a scheduler writes the device word)."""
    _need_cuda(p, g, m, v)
    st = _stream()
    if step_dev is not None and advance:     # (advance = False: a second launch of the same optimizer step over another slice)
        lib.call("avid_counter_add", _p(step_dev), 1, st)
    lib.call("avid_adam_flat", p.numel(), _p(p), _p(g), _p(m), _p(v), float(lr), float(beta1), float(beta2),
             float(eps), float(wd), int(step), _p(step_dev), _p(lr_dev), float(grad_scale), st)
