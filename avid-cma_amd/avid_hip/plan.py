"""Launch programs for the two-tower model: the whole forward / backward pass of ``AV_Wrapper`` as ONE C call each.

The reference drives its step from Python, one ATen call per layer (main-avid.py:167,179 -> models/av_wrapper.py:50-61
-> models/video.py:44-54, models/audio.py:33-44, models/network_blocks.py:23-27,52-60).  Doing the same with this
library's entry points (``avid_hip.ops``: one ``torch.autograd.Function`` per layer) costs ~330 launches per step through
the interpreter — 7.7 ms of host time against 11.2 ms of GPU time in round 3, and the reference's own loop, which
synchronises on ``loss.item()`` every step, saw all of it.  Here the launch sequence is COMPILED once per (model, input
shape, stream arrangement) into an ``avid_instr`` array (include/avid_hip.h, "Launch programs") and executed by
``avid_program_run`` in C: same entry points, same arguments, same order per stream as the per-layer path — bit-identical
results (tests/test_gpu_plan.py) — and the whole model is one autograd node.

* ``Builder`` walks the module tree (R2Plus1D / Conv2D / Head, i.e. models/*.py of this package) and emits the forward
  records, then the backward records in autograd's order, with every fusion decision of the per-layer path taken at
  compile time: BatchNorm statistics from the convolution epilogue, BatchNorm-backward sums from the consumer's
  input-gradient kernel, residual add in the epilogue, compact gradient of the strided 1x1x1 residual convolution,
  grouped weight gradients, weight gradients on trailing streams, the audio tower on its own stream.
* tensors are ``(slot, byte offset)`` references: activations live in ONE arena per pass (a torch tensor the autograd
  node keeps alive), parameters / buffers / inputs are slots re-read from the tensors on every run — the program holds
  no address, so re-seated parameters (``FlatParams``), a new batch or a different allocator block just work.
* ``NetFn`` is the autograd node; parameter gradients go straight into a flat gradient buffer (the step engine's, or a
  fresh one whose slices are handed to autograd so that ``torch.optim.Adam`` / DDP see ordinary ``.grad`` tensors).
"""
from __future__ import annotations

import ctypes as C
import os
import struct

import torch

from . import lib, ops
from .lib import ConvDesc

# ---- ctypes mirrors of include/avid_hip.h ------------------------------------------------------------------------
OP_WAIT, OP_MEMSET0, OP_CONV_FWD, OP_CONV_DGRAD, OP_CONV_WGRAD, OP_WGRAD_GROUP, OP_WGRAD_ITEM = 1, 2, 3, 4, 5, 6, 7
OP_BN_FWD, OP_BN_BWD, OP_BN_POOL_FWD, OP_BN_POOL_BWD, OP_GPOOL_FWD, OP_GPOOL_BWD = 8, 9, 10, 11, 12, 13
OP_RELU_BWD, OP_COLSUM, OP_WT_BATCH, OP_ADAM = 14, 15, 16, 17
NREF = lib.INSTR_REFS
Ref, Instr, StreamWs = lib.Ref, lib.Instr, lib.StreamWs
_vp, _i, _sz = C.c_void_p, C.c_int, C.c_size_t

# ---- slots ---------------------------------------------------------------------------------------------------------
S_FWD, S_BWD, S_GRAD, S_AUX, S_VIDEO, S_AUDIO, S_DV, S_DA, S_FIRST_TENSOR = 0, 1, 2, 3, 4, 5, 6, 7, 8
NULL = (-1, 0)
# streams of a run (avid_hip/streams.py: one per dispatch pipe): compute (video tower), audio tower, the trailing
# weight-gradient stream of both, the collectives' stream (not used by the programs themselves)
ST_MAIN, ST_AUDIO, ST_TRAIL, ST_COMM = 0, 1, 2, 3

ENABLED = os.environ.get("AVID_PLAN", "1") == "1"


class Unsupported(Exception):
    """The module tree holds something the compiler does not know: the caller falls back to the per-layer path."""


def _align(n, a=256):
    return (n + a - 1) // a * a


class Arena:
    """Bump allocator of one pass (no reuse: a trailing weight gradient may still read a tensor long after the chain
    has moved on, and 288 GB of HBM hold the ~3 GB per pass of a 64-clip batch many times over)."""

    def __init__(self, slot):
        self.slot, self.size = slot, 0

    def alloc(self, nbytes):
        off = self.size
        self.size += _align(max(int(nbytes), 4))
        return (self.slot, off)


class Sym:
    """A tensor known to the compiler: reference, channels-last shape, and — for the output of a training-mode
    BatchNorm(+ReLU) whose only consumer is the next convolution — that BatchNorm's record (``ops.BnSource``)."""
    __slots__ = ("ref", "shape", "bn", "affine")

    def __init__(self, ref, shape, bn=None, affine=None):
        """``affine`` = (reference of the BatchNorm's saved [4][C] vectors, relu, C): ``ref`` is then the BatchNorm's INPUT and
        the tensor this Sym stands for — ReLU(bn(ref)) — exists nowhere: its consumers apply the map themselves
        (OP_CONV_FWD's i1 / OP_CONV_WGRAD's i0)."""
        self.ref, self.shape, self.bn, self.affine = ref, tuple(shape), bn, affine

    @property
    def numel(self):
        n = 1
        for v in self.shape:
            n *= v
        return n


class BnRec:
    __slots__ = ("x", "s4", "relu", "C", "partials", "rows")

    def __init__(self, x, s4, relu, Cc):
        self.x, self.s4, self.relu, self.C, self.partials, self.rows = x, s4, relu, Cc, None, 0


def _off(ref, nbytes):
    return (ref[0], ref[1] + nbytes)


class Builder:
    def __init__(self, device, overlap_towers, trailing, group):
        self.device = device
        self.overlap, self.trailing, self.group = overlap_towers, trailing, group
        self.fa, self.ba = Arena(S_FWD), Arena(S_BWD)
        self.fwd, self.bwd = [], []
        self.cur = self.fwd
        self.tensors, self.tslot = [], {}          # external tensors (parameters, buffers) -> slot
        self.params, self.pindex = [], {}          # trainable parameters in gradient-buffer order
        self.S = ST_MAIN                           # compute stream of the records being emitted
        self.pending = {ST_MAIN: [], ST_AUDIO: []}  # queued weight gradients per compute stream
        self.trail_used = set()
        self.tables_waited = set()                 # streams whose forward has waited for the weight-transform launch
        self.wt, self.wt_off = {}, {}              # transposed weights / Winograd transforms: param slot -> aux offset
        self.aux_size = 0
        self.wt_recs = []                          # (param tensor, aux offset, Cout, taps, Cin, mode)
        self.grad_ready = []                       # (index of the backward record that completes them, [param index])

    # ---- references
    def ext(self, t):
        """Slot of an external tensor (parameter / buffer), registered on first use."""
        key = id(t)
        s = self.tslot.get(key)
        if s is None:
            s = self.tslot[key] = S_FIRST_TENSOR + len(self.tensors)
            self.tensors.append(t)
        return (s, 0)

    def grad(self, p):
        """Reference of parameter p's slice of the flat gradient buffer."""
        return (S_GRAD, 4 * self.goff[self.pindex[id(p)]])

    def set_params(self, params):
        """Gradient-buffer layout = ``parallel.FlatParams``': reverse registration order, 16-byte aligned slices."""
        self.params = list(reversed(params))
        self.goff, off = [], 0
        for k, p in enumerate(self.params):
            self.pindex[id(p)] = k
            self.goff.append(off)
            off += (p.numel() + 3) // 4 * 4
        self.gnumel = off

    def emit(self, op, stream=None, d=None, i=(), n=(), f=(), t=()):
        ins = Instr()
        ins.op = op
        ins.stream = self.S if stream is None else stream
        if d is not None:
            ins.d = d
        for k, v in enumerate(i):
            ins.i[k] = int(v)
        for k, v in enumerate(n):
            ins.n[k] = int(v)
        for k, v in enumerate(f):
            ins.f[k] = float(v)
        for k in range(NREF):
            r = t[k] if k < len(t) else NULL
            if r is None:
                r = NULL
            ins.t[k].slot, ins.t[k].off = int(r[0]), int(r[1])
        self.cur.append(ins)
        return ins

    def wait(self, waiter, waited):
        if waiter != waited:
            self.emit(OP_WAIT, stream=0, i=(waiter, waited))

    def trail(self):
        return ST_TRAIL

    # ---- transposed weights / Winograd transforms for the input gradients (ops.TransposedWeights)
    def want_wt(self, w):
        key = (id(w), 0)
        if key not in self.wt_off:
            if not (w.shape[1] % 64 == 0 and w.shape[0] % 32 == 0):
                return None
            k = ops._kdims(w)
            self.wt_off[key] = self.aux_size
            self.wt_recs.append((w, self.aux_size, w.shape[0], k[0] * k[1] * k[2], w.shape[1], 0))
            self.aux_size += _align(4 * w.numel())
        return (S_AUX, self.wt_off[key])

    def want_u(self, w, variant, forward=False):
        code = (1 if forward else 2) + 2 * (variant - 1)   # forward (1 / 3) or input gradient (2 / 4), wino_kernel's / wino2_kernel's order
        key = (id(w), code)
        if key not in self.wt_off:
            self.wt_off[key] = self.aux_size
            self.wt_recs.append((w, self.aux_size, w.shape[0], 9, w.shape[1], code))
            self.aux_size += _align(4 * ops.U_FLOATS * w.shape[0] * w.shape[1])
        return (S_AUX, self.wt_off[key])

    def want_split(self, w, mode):
        """The weights pre-split into bf16 terms for igemm_pk_kernel's 128 x 64 tile (mode 5: forward, 6: input gradient)."""
        key = (id(w), mode)
        if key not in self.wt_off:
            k = ops._kdims(w)
            self.wt_off[key] = self.aux_size
            self.wt_recs.append((w, self.aux_size, w.shape[0], k[0] * k[1] * k[2], w.shape[1], mode))
            self.aux_size += _align(6 * w.numel())
            if mode == 5:
                self.fwd_tables = True
        return (S_AUX, self.wt_off[key])

    def fwd_u(self, w, d):
        """`u` of a forward convolution on stream self.S.  The pre-split weights (want_split mode 5) come out of the step's
        weight-transform launch on the trailing stream, which is issued beside the video stem and starves behind its
        persistent workgroups (parallel.py: forward_backward): the FIRST layer of a stream that reads a table waits for
        that launch — on the video tower that is conv2x's temporal convolution, 0.2 ms behind the stem, and costs nothing;
        a wait right behind the stem stalled the compute stream for the time the tables were late (10.07-10.3 against
        10.03-10.09 ms per step on one box)."""
        if d.wino_fwd:
            # a Winograd layer behind that wait takes its transformed weights from the same launch (mode 1 / 3) instead of
            # transforming them in front of its own kernel (one 6 us launch per layer on the compute stream); the first
            # one of the video tower sits right behind the stem and keeps its own
            if self.trailing and self.S in self.tables_waited:
                return self.want_u(w, d.wino_fwd, forward=True)
            return None
        if not d.split_fwd:
            return None
        if self.trailing and self.S not in self.tables_waited:
            self.wait(self.S, ST_TRAIL)
            self.tables_waited.add(self.S)
        return self.want_split(w, 5)

    def dgrad_wt_u(self, w, d):
        """(wt, u) of an input gradient: the Winograd transform, the pre-split transpose, or the plain transpose."""
        if d.wino_dgrad:
            return self.want_wt(w), self.want_u(w, d.wino_dgrad)
        if d.split_dgrad:
            return None, self.want_split(w, 6)
        return self.want_wt(w), None

    # ---- forward emitters ------------------------------------------------------------------------------------------
    def desc(self, x_shape, conv_w, stride, pad, channel_first):
        if channel_first:
            B, c, Ti, Hi, Wi = x_shape
        else:
            B, Ti, Hi, Wi, c = x_shape
        cout, cin = conv_w.shape[0], conv_w.shape[1]
        if c != cin:
            raise Unsupported("channel mismatch")
        if not ops.weight_layout_ok(conv_w) or conv_w.dtype != torch.float32:
            raise Unsupported("weight layout")
        return ops._desc_cached((B, Ti, Hi, Wi), cin, cout, ops._kdims(conv_w), tuple(stride), tuple(pad), channel_first)

    def conv_bn_fwd(self, conv, bn, x, addend=None, res=None, sole=True, next_conv=None):
        """ReLU(bn(conv(x) [+ addend])) (network_blocks._conv_bn in training mode).  Returns the layer record.

        ``next_conv``: the ONLY consumer of this layer's output, if that is known to be a convolution.  When its kernels can apply
        a BatchNorm (+ReLU) to their input while they stage it (``d.in_affine``: conv2x's temporal layers), this layer's
        BatchNorm makes its statistics only and the returned ``h`` is the convolution's OUTPUT tagged with the map
        (``Sym.affine``): the normalised tensor is never written or read."""
        w = conv.weight
        d, _, _, _, srows = self.desc(x.shape, w, conv.stride3, conv.padding3, conv.channel_first)
        ysh = (d.B, d.To, d.Ho, d.Wo, d.Cout)
        M = d.B * d.To * d.Ho * d.Wo
        y = self.fa.alloc(4 * M * d.Cout)
        stats = self.fa.alloc(4 * srows * 2 * d.Cout) if srows > 0 else None
        xa = x.affine
        if xa is not None and not (d.in_affine and res is None):
            raise Unsupported("a tensor with a pending BatchNorm reached a layer that cannot apply it")
        self.emit(OP_CONV_FWD, d=d, i=(0,) if xa is None else (0, 2 if xa[1] else 1, xa[2]),
                  t=(x.ref, self.ext(w), self.fwd_u(w, d), addend.ref if addend is not None else None, None, y, stats) +
                    (() if xa is None else (xa[0],)))
        L = {"conv": conv, "bn": bn, "d": d, "x": x, "y": y, "w": w, "sole": sole, "res": None, "M": M}
        if res is not None:
            rconv = res
            rw = rconv.weight
            if conv.channel_first or ops._kdims(rw) != (1, 1, 1) or rw.shape[1] != d.Cin:
                raise Unsupported("residual convolution")
            dr = self.desc(x.shape, rw, rconv.stride3, (0, 0, 0), False)[0]
            y_res = self.fa.alloc(4 * d.B * dr.To * dr.Ho * dr.Wo * dr.Cout)
            self.emit(OP_CONV_FWD, d=dr, i=(0,), t=(x.ref, self.ext(rw), self.fwd_u(rw, dr), None, None, y_res, None))
            L["res"] = {"conv": rconv, "d": dr, "w": rw, "y": Sym(y_res, (d.B, dr.To, dr.Ho, dr.Wo, dr.Cout))}
        Cc = d.Cout
        defer = False
        if next_conv is not None and addend is None and res is None:
            nw = next_conv.weight
            if ops.weight_layout_ok(nw) and nw.dtype == torch.float32 and nw.shape[1] == Cc and not next_conv.channel_first:
                defer = self.desc(ysh, nw, next_conv.stride3, next_conv.padding3, False)[0].in_affine
        h = None if defer else self.fa.alloc(4 * M * Cc)
        s4 = self.fa.alloc(4 * 4 * Cc)
        self.emit(OP_BN_FWD, n=(M,), i=(Cc, 1, srows if stats is not None else 0), f=(bn.momentum, bn.eps),
                  t=(y, self.ext(bn.weight), self.ext(bn.bias), self.ext(bn.running_mean), self.ext(bn.running_var), h, s4,
                     self.ext(bn.num_batches_tracked), stats))
        rec = BnRec(y, s4, 1, Cc)
        L["bnrec"] = rec
        L["h"] = Sym(y, ysh, bn=rec, affine=(s4, 1, Cc)) if defer else Sym(h, ysh, bn=rec)
        return L

    def linear_fwd(self, lin, x, relu):
        """nn.Linear (+ReLU) as a 1x1x1 convolution over [B,1,1,1,C] (ops.linear)."""
        w = lin.weight
        B, Cin = x.shape
        d = ops._desc_cached((B, 1, 1, 1), w.shape[1], w.shape[0], (1, 1, 1), (1, 1, 1), (0, 0, 0), False)[0]
        if Cin != w.shape[1] or not ops.weight_layout_ok(w):
            raise Unsupported("linear")
        y = self.fa.alloc(4 * B * d.Cout)
        self.emit(OP_CONV_FWD, d=d, i=(1 if relu else 0,), t=(x.ref, self.ext(w), None, None, self.ext(lin.bias), y, None))
        return {"lin": lin, "d": d, "x": x, "y": Sym(y, (B, d.Cout)), "relu": relu, "w": w}

    def gpool_fwd(self, x):
        B, Cc = x.shape[0], x.shape[-1]
        S = x.numel // (B * Cc)
        y = self.fa.alloc(4 * B * Cc)
        am = self.fa.alloc(4 * B * Cc)
        self.emit(OP_GPOOL_FWD, i=(B, S, Cc), t=(x.ref, y, am))
        return {"x": x, "y": Sym(y, (B, Cc)), "am": am, "S": S}

    # ---- backward emitters -----------------------------------------------------------------------------------------
    def _ready(self, stream, *ps):
        """The records emitted so far complete the gradients of ``ps`` (written on ``stream``)."""
        self.grad_ready.append((len(self.bwd), stream, [self.pindex[id(p)] for p in ps]))

    def flush_group(self, S=None):
        S = self.S if S is None else S
        lst, self.pending[S] = self.pending[S], []
        if not lst:
            return
        stream = S
        if self.trailing:
            stream = ST_TRAIL
            self.wait(stream, S)                      # every dy of the group is complete on its compute stream
            self.trail_used.add(stream)
        self.emit(OP_WGRAD_GROUP, stream=stream, i=(len(lst),))
        for d, xr, dyr, w in lst:
            self.emit(OP_WGRAD_ITEM, stream=stream, d=d, t=(xr, dyr, self.grad(w)))
        self._ready(stream, *[t[3] for t in lst])

    def wgrad(self, d, xr, dyr, w, flush_check=True, xa=None):
        """The weight gradient of one layer: queued for a grouped launch (small layers), or its own launch on the
        trailing stream; False = the caller emits it on the compute stream (behind the input gradient)."""
        if self.group and d.groupable:
            if xa is not None:
                raise Unsupported("a grouped weight gradient cannot apply its input's BatchNorm")
            self.pending[self.S].append((d, xr, dyr, w))
            if len(self.pending[self.S]) >= ops.GROUP_MAX:
                self.flush_group()
            return True
        if flush_check and self.group and len(self.pending[self.S]) >= ops.GROUP_MIN_FLUSH:
            self.flush_group()                        # the small layers queued so far go first
        if self.trailing:
            tr = self.trail()
            self.wait(tr, self.S)
            self.emit(OP_CONV_WGRAD, stream=tr, d=d, **self._wgrad_args(xr, dyr, w, xa))
            self._ready(tr, w)
            self.trail_used.add(tr)
            return True
        return False                                   # the caller emits it behind the input gradient

    def _wgrad_args(self, xr, dyr, w, xa):
        """i / t of an OP_CONV_WGRAD record; xa = Sym.affine of the layer's input (None: x is read as it is)."""
        if xa is None:
            return {"t": (xr, dyr, self.grad(w))}
        return {"i": (2 if xa[1] else 1, xa[2]), "t": (xr, dyr, self.grad(w), xa[0])}

    def wgrad_inline(self, d, xr, dyr, w, xa=None):
        self.emit(OP_CONV_WGRAD, d=d, **self._wgrad_args(xr, dyr, w, xa))
        self._ready(self.S, w)

    def conv_bwd(self, d, w, x, g, need_dx, add=None, add_stride=None, res=None, g_res=None):
        """Backward of one convolution given g = d(loss)/d(conv output): weight gradient, [residual branch,] input
        gradient (ops._ConvCL.backward, same order of launches per stream).  Returns the reference of dx."""
        done = self.wgrad(d, x.ref, g, w, xa=x.affine)
        if res is not None:
            if x.affine is not None:
                raise Unsupported("residual convolution on a tensor with a pending BatchNorm")
            dr, rw = res["d"], res["w"]
            if need_dx:
                # compact input gradient of the residual convolution: a dense 1x1x1 dgrad over the sub-sampled grid
                dc = ops._desc_cached((d.B, dr.To, dr.Ho, dr.Wo), d.Cin, dr.Cout, (1, 1, 1), (1, 1, 1), (0, 0, 0), False)[0]
                add = self.ba.alloc(4 * d.B * dr.To * dr.Ho * dr.Wo * d.Cin)
                self.emit(OP_CONV_DGRAD, d=dc, i=(0, 0, 0, 0, 0), t=(g_res, self.ext(rw), *self.dgrad_wt_u(rw, dc), None, add))
                rs = res["conv"].stride3
                add_stride = tuple(rs) if any(v != 1 for v in rs) else None
            if not self.wgrad(dr, x.ref, g_res, rw, flush_check=False):
                self.wgrad_inline(dr, x.ref, g_res, rw)
        dx = None
        if need_dx:
            dx = self.ba.alloc(4 * x.numel)
            src = x.bn
            fuse = src is not None and ops.FUSE_BN_BWD and d.bn_bwd_rows > 0
            t = [g, self.ext(w), *self.dgrad_wt_u(w, d), add, dx]
            iv = list(add_stride) if add_stride is not None else [0, 0, 0]
            if fuse:
                src.rows = d.bn_bwd_rows
                src.partials = self.ba.alloc(4 * src.rows * 2 * d.Cin)
                Cc = src.C
                t += [src.x, _off(src.s4, 8 * Cc), _off(src.s4, 12 * Cc), src.s4, _off(src.s4, 4 * Cc), src.partials]
                iv += [src.relu, 1]
            else:
                iv += [0, 0]
            self.emit(OP_CONV_DGRAD, d=d, i=iv, t=t)
        if not done:
            self.wgrad_inline(d, x.ref, g, w, xa=x.affine)
        return dx

    def conv_bn_bwd(self, L, dh, need_dx=True, add=None, g_res=None):
        """Backward of a conv_bn_fwd layer given dh = d(loss)/d(h).  Returns (g, dx): g = gradient at the convolution's
        output (== gradient of its addend), dx = gradient of its input (None if not needed)."""
        d, bn, rec = L["d"], L["bn"], L["bnrec"]
        g = self.ba.alloc(4 * L["M"] * d.Cout)
        self.emit(OP_BN_BWD, n=(L["M"],), i=(d.Cout, 1, rec.rows if rec.partials is not None else 0, 0),
                  t=(L["y"], dh, self.ext(bn.weight), rec.s4, g, self.grad(bn.weight), self.grad(bn.bias), rec.partials))
        self._ready(self.S, bn.weight, bn.bias)
        x = L["x"]
        if not L["sole"]:
            x = Sym(x.ref, x.shape, affine=x.affine)  # not the only consumer: no BatchNorm hand-over
        dx = self.conv_bwd(d, L["w"], x, g, need_dx, add=add, res=L["res"], g_res=g_res)
        return g, dx

    def linear_bwd(self, L, dy):
        d, lin = L["d"], L["lin"]
        B = d.B
        if L["relu"]:
            g = self.ba.alloc(4 * B * d.Cout)
            self.emit(OP_RELU_BWD, n=(B * d.Cout,), t=(L["y"].ref, dy, g))
            dy = g
        dx = self.conv_bwd(d, L["w"], L["x"], dy, True)
        self.emit(OP_COLSUM, n=(B,), i=(d.Cout,), t=(dy, self.grad(lin.bias)))
        self._ready(self.S, lin.bias)
        return dx

    def gpool_bwd(self, P, dy):
        x = P["x"]
        dx = self.ba.alloc(4 * x.numel)
        self.emit(OP_GPOOL_BWD, i=(x.shape[0], P["S"], x.shape[-1]), t=(dy, P["am"], dx))
        return dx

    # ---- module walkers --------------------------------------------------------------------------------------------
    def r2p1d_block_fwd(self, blk, x):
        from models.network_blocks import BasicR2P1DBlock
        if type(blk) is not BasicR2P1DBlock:
            raise Unsupported(type(blk).__name__)
        if blk.res:
            from models import network_blocks as nb
            rs, ss = blk.res_conv.stride3, blk.spt_conv1.stride3
            if not (nb._FUSE_RES and any(v == 2 for v in rs) and all(v in (1, 2) for v in rs)
                    and any(v == 2 for v in ss) and all(v in (1, 2) for v in ss) and x.numel * 4 < (1 << 31)):
                raise Unsupported("residual convolution outside the fused pattern")
        # (spt_bn1 -> tmp_conv1 and spt_bn2 -> tmp_conv2: each BatchNorm's output has exactly one reader, the next convolution)
        L1 = self.conv_bn_fwd(blk.spt_conv1, blk.spt_bn1, x, res=blk.res_conv if blk.res else None,
                              next_conv=None if blk.res else blk.tmp_conv1)
        L2 = self.conv_bn_fwd(blk.tmp_conv1, blk.tmp_bn1, L1["h"])
        L3 = self.conv_bn_fwd(blk.spt_conv2, blk.spt_bn2, L2["h"], next_conv=blk.tmp_conv2)
        addend = L1["res"]["y"] if blk.res else x
        L4 = self.conv_bn_fwd(blk.tmp_conv2, blk.out_bn, L3["h"], addend=addend)
        return (L1, L2, L3, L4), L4["h"]

    def r2p1d_block_bwd(self, Ls, dout):
        L1, L2, L3, L4 = Ls
        g4, dh3 = self.conv_bn_bwd(L4, dout)
        _, dh2 = self.conv_bn_bwd(L3, dh3)
        _, dh1 = self.conv_bn_bwd(L2, dh2)
        if L1["res"] is not None:
            _, dx = self.conv_bn_bwd(L1, dh1, g_res=g4)
        else:
            _, dx = self.conv_bn_bwd(L1, dh1, add=g4)       # identity residual: its gradient rides in the dgrad's addend
        return dx

    def video_fwd(self, vm, x, after_stem=None):
        from models.video import R2Plus1D
        from models.network_blocks import BatchNormCL, ConvCL
        if type(vm) is not R2Plus1D:
            raise Unsupported(type(vm).__name__)
        conv, bn = vm.conv1[0], vm.conv1[1]
        if type(conv) is not ConvCL or type(bn) is not BatchNormCL:
            raise Unsupported("video stem")
        w = conv.weight
        d, _, _, _, srows = self.desc(x.shape, w, conv.stride3, conv.padding3, True)
        M = d.B * d.To * d.Ho * d.Wo
        y = self.fa.alloc(4 * M * d.Cout)
        stats = self.fa.alloc(4 * srows * 2 * d.Cout) if srows > 0 else None
        self.emit(OP_CONV_FWD, d=d, i=(0,), t=(x.ref, self.ext(w), None, None, None, y, stats))
        B, T, H, W, Cc = d.B, d.To, d.Ho, d.Wo, d.Cout
        Ho, Wo = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
        p = self.fa.alloc(4 * B * T * Ho * Wo * Cc)
        am = self.fa.alloc(B * T * Ho * Wo * Cc)
        s4 = self.fa.alloc(16 * Cc)
        self.emit(OP_BN_POOL_FWD, i=(B, T, H, W, Cc, srows if stats is not None else 0), f=(bn.momentum, bn.eps),
                  t=(y, self.ext(bn.weight), self.ext(bn.bias), self.ext(bn.running_mean), self.ext(bn.running_var), p, am, s4,
                     self.ext(bn.num_batches_tracked), stats))
        stem = {"d": d, "w": w, "x": x, "y": y, "am": am, "s4": s4, "bn": bn, "dims": (B, T, H, W, Cc)}
        h = Sym(p, (B, T, Ho, Wo, Cc))
        if after_stem is not None:
            after_stem()
        blocks = []
        for name in ("conv2x", "conv3x", "conv4x", "conv5x"):
            stage = getattr(vm, name)
            for blk in (stage if isinstance(stage, torch.nn.Sequential) else [stage]):
                Ls, h = self.r2p1d_block_fwd(blk, h)
                blocks.append(Ls)
        pool = self.gpool_fwd(h)
        return {"stem": stem, "blocks": blocks, "pool": pool}, pool["y"]

    def video_bwd(self, V, dy):
        dh = self.gpool_bwd(V["pool"], dy)
        for Ls in reversed(V["blocks"]):
            dh = self.r2p1d_block_bwd(Ls, dh)
        return dh

    def video_stem_bwd(self, V, dh):
        st = V["stem"]
        B, T, H, W, Cc = st["dims"]
        bn = st["bn"]
        dx = self.ba.alloc(4 * B * T * H * W * Cc)
        self.emit(OP_BN_POOL_BWD, i=(B, T, H, W, Cc),
                  t=(st["y"], dh, st["am"], self.ext(bn.weight), st["s4"], dx, self.grad(bn.weight), self.grad(bn.bias)))
        self._ready(self.S, bn.weight, bn.bias)
        self.conv_bwd(st["d"], st["w"], st["x"], dx, False)

    def audio_fwd(self, am, x):
        from models.audio import Conv2D
        from models.network_blocks import Basic2DBlock
        if type(am) is not Conv2D:
            raise Unsupported(type(am).__name__)
        Ls = [self.conv_bn_fwd(am.conv1[0], am.conv1[1], x)]
        h = Ls[0]["h"]
        for blk in (am.block1, am.block2, am.block3, am.block4):
            if type(blk) is not Basic2DBlock:
                raise Unsupported(type(blk).__name__)
            for conv, bn in ((blk.conv1, blk.bn1), (blk.conv2, blk.bn2)):
                L = self.conv_bn_fwd(conv, bn, h)
                Ls.append(L)
                h = L["h"]
        pool = self.gpool_fwd(h)
        return {"layers": Ls, "pool": pool}, pool["y"]

    def audio_bwd(self, A, dy):
        dh = self.gpool_bwd(A["pool"], dy)
        Ls = A["layers"]
        for k in range(len(Ls) - 1, -1, -1):
            _, dh = self.conv_bn_bwd(Ls[k], dh, need_dx=k > 0)

    def head_fwd(self, head, x):
        from models.av_wrapper import Head, LinearCL
        if type(head) is not Head:
            raise Unsupported(type(head).__name__)
        mods = list(head.projection)
        Ls, k = [], 0
        while k < len(mods):
            if type(mods[k]) is not LinearCL:
                raise Unsupported("head")
            relu = k + 1 < len(mods) and isinstance(mods[k + 1], torch.nn.ReLU)
            L = self.linear_fwd(mods[k], x, relu)
            Ls.append(L)
            x = L["y"]
            k += 2 if relu else 1
        return Ls, x

    def head_bwd(self, Ls, dy):
        for L in reversed(Ls):
            dy = self.linear_bwd(L, dy)
        return dy


def _hooked(model):
    for m in model.modules():
        if m._forward_hooks or m._forward_pre_hooks or m._backward_hooks or m._backward_pre_hooks:
            return True
    return False


class Plan:
    """The compiled forward / backward programs of one ``AV_Wrapper`` for one input geometry and stream arrangement."""

    def __init__(self, model, vshape, ashape, device, overlap_towers, trailing, group):
        if not model.use_linear_proj:
            raise Unsupported("no projection heads")
        params = [p for p in model.parameters()]
        dry = device.type != "cuda"                  # (compile only: tests/test_plan_compile.py, no GPU)
        if not all(p.requires_grad and (p.is_cuda or dry) and p.dtype == torch.float32 for p in params):
            raise Unsupported("frozen / non-fp32 parameters")
        b = Builder(device, overlap_towers, trailing, group)
        b.set_params(params)
        self.device = device
        self.key = (tuple(vshape), tuple(ashape), overlap_towers, trailing, group)
        B = vshape[0]
        video = Sym((S_VIDEO, 0), vshape)
        audio = Sym((S_AUDIO, 0), (ashape[0], ashape[1], 1, ashape[2], ashape[3]))   # [B,1,1,H,W]: T = kt = 1
        A_S = ST_AUDIO if overlap_towers else ST_MAIN
        # ------------------------------------------------------------------ forward
        b.cur = b.fwd
        # (what the backward needs but the forward does not — the transposed / Winograd-transformed weights — runs on
        #  the video tower's trailing stream beside the forward; the table is known only after the backward is
        #  compiled, so the record is patched in below)
        helper = ST_TRAIL if trailing else ST_MAIN
        if trailing:
            b.wait(helper, ST_MAIN)
        self._zero_index = len(b.fwd)
        b.emit(OP_MEMSET0, stream=helper, n=(4 * b.gnumel,), t=((S_GRAD, 0),))
        self._wt_rec = b.emit(OP_WT_BATCH, stream=helper, i=(0,), n=(0,), t=((S_AUX, 0),))
        b.S = ST_MAIN
        # the video stem first, then the audio tower starts on its own stream (beside conv2x.., not beside the stem:
        # models/av_wrapper.py AUDIO_AFTER)
        V, vfeat = self._video_with_audio(b, model, video, audio, A_S)
        A, afeat = self._A, self._afeat
        b.S = ST_MAIN
        VH, vemb = b.head_fwd(model.video_proj, vfeat)
        if A_S != ST_MAIN:
            b.wait(ST_MAIN, A_S)
        self.vemb, self.aemb = vemb, self._aemb
        self.n_fwd = len(b.fwd)
        # ------------------------------------------------------------------ backward (autograd's order: the video
        # tower from its head down to conv2x, the audio tower, the video stem last — sequence numbers of the forward)
        b.cur = b.bwd
        if trailing:
            b.wait(ST_MAIN, helper)                   # the transposed weights are complete
        if A_S != ST_MAIN:
            b.wait(A_S, ST_MAIN)                      # the criterion's gradients (and the transposed weights)
        b.S = ST_MAIN
        dv = b.head_bwd(VH, (S_DV, 0))
        dh = b.video_bwd(V, dv)
        b.S = A_S
        da = b.head_bwd(self._AH, (S_DA, 0))
        b.audio_bwd(A, da)
        # Everything but the video stem's three parameters has its gradient here: the engine's optimizer can update those
        # (nearly all of the 134 MB) on the fourth stream WHILE the stem's backward — one LDS-bound full-chip kernel of
        # 0.57 ms on the trailing stream, the step's last — still runs (parallel.TrainStep.step; the HBM-bound Adam pass and
        # that kernel overlap).  The fourth stream waits here for the three streams that hold those gradients.
        for S in (ST_MAIN, ST_AUDIO):
            b.S = S
            b.flush_group(S)
        b.S = ST_MAIN
        self.adam_early = None
        if trailing and A_S != ST_MAIN:
            stem_conv, stem_bn = model.video_model.conv1[0], model.video_model.conv1[1]
            idx = sorted(b.pindex[id(p)] for p in (stem_conv.weight, stem_bn.weight, stem_bn.bias))
            n = len(b.params)
            if idx == [n - 3, n - 2, n - 1]:                   # (the engine's flat layout: parameters in reverse order)
                for S in (ST_MAIN, A_S, ST_TRAIL):
                    b.wait(ST_COMM, S)
                self.adam_early = b.goff[n - 3]
        b.video_stem_bwd(V, dh)
        b.flush_group(ST_MAIN)
        if A_S != ST_MAIN:
            b.wait(ST_MAIN, A_S)
        for tr in sorted(b.trail_used):
            b.wait(ST_MAIN, tr)
        self.n_bwd = len(b.bwd)
        # ------------------------------------------------------------------ the weight-transform table
        self.aux_bytes = _align(b.aux_size) + 32 * len(b.wt_recs) + 256
        self.table_off = _align(b.aux_size)
        self.wt_recs = b.wt_recs
        self._wt_rec.i[0] = len(b.wt_recs)
        self._wt_rec.n[0] = max([r[2] * r[3] * r[4] for r in b.wt_recs] + [1])
        self._wt_rec.t[0].off = self.table_off
        if not b.wt_recs:
            self._wt_rec.op = 0
        self.fwd_prog = (Instr * max(1, self.n_fwd))(*b.fwd)
        self.bwd_prog = (Instr * max(1, self.n_bwd))(*b.bwd)
        self.fa_bytes, self.ba_bytes = max(b.fa.size, 256), max(b.ba.size, 256)
        self.params, self.goff, self.gnumel = b.params, b.goff, b.gnumel
        self.params_fwd = params
        self.tensors = b.tensors
        self.n_slots = S_FIRST_TENSOR + len(self.tensors)
        self.grad_ready = b.grad_ready
        self.n_streams = 4
        need_f, need_b = (_sz * 4)(), (_sz * 4)()
        lib.call("avid_program_workspace_bytes", self.fwd_prog, 0, self.n_fwd, 4, need_f)
        lib.call("avid_program_workspace_bytes", self.bwd_prog, 0, self.n_bwd, 4, need_b)
        self.ws_bytes = [max(int(need_f[k]), int(need_b[k]), 1 << 20) for k in range(4)]
        self.ws = [torch.empty(n, dtype=torch.uint8, device=device) for n in self.ws_bytes]
        self.ws_arr = (StreamWs * 4)()
        for k in range(4):
            self.ws_arr[k].ptr, self.ws_arr[k].bytes = self.ws[k].data_ptr(), self.ws_bytes[k]
        self.aux = torch.empty(self.aux_bytes, dtype=torch.uint8, device=device)
        self._table_ptrs = None
        self.slots = (_vp * self.n_slots)()
        self.streams = (_vp * 4)()
        self.vshape, self.ashape = tuple(vshape), tuple(ashape)

    def _video_with_audio(self, b, model, video, audio, A_S):
        """The video tower with the audio tower (and its head) started behind the video stem on stream A_S."""
        def start_audio():
            S0 = b.S
            if A_S != ST_MAIN:
                b.wait(A_S, ST_MAIN)
            b.S = A_S
            self._A, self._afeat = b.audio_fwd(model.audio_model, audio)
            self._AH, self._aemb = b.head_fwd(model.audio_proj, self._afeat)
            b.S = S0
        return b.video_fwd(model.video_model, video, after_stem=start_audio)

    # ---- per-run plumbing ------------------------------------------------------------------------------------------
    def _fill_slots(self, fa, ba, grad, video, audio, dv, da):
        s = self.slots
        s[S_FWD] = fa.data_ptr() if fa is not None else None
        s[S_BWD] = ba.data_ptr() if ba is not None else None
        s[S_GRAD] = grad.data_ptr() if grad is not None else None
        s[S_AUX] = self.aux.data_ptr()
        s[S_VIDEO] = video.data_ptr() if video is not None else None
        s[S_AUDIO] = audio.data_ptr() if audio is not None else None
        s[S_DV] = dv.data_ptr() if dv is not None else None
        s[S_DA] = da.data_ptr() if da is not None else None
        ptrs = [t.data_ptr() for t in self.tensors]
        s[S_FIRST_TENSOR:self.n_slots] = ptrs
        return ptrs

    def _refresh_table(self):
        """The device table of weight-transform descriptors follows the parameters' current addresses."""
        ptrs = tuple(r[0].data_ptr() for r in self.wt_recs)
        if ptrs != self._table_ptrs and self.wt_recs:
            base = self.aux.data_ptr()
            recs = b"".join(struct.pack("<QQiiii", p, base + r[1], r[2], r[3], r[4], r[5]) for p, r in zip(ptrs, self.wt_recs))
            host = torch.frombuffer(bytearray(recs), dtype=torch.uint8)
            self.aux[self.table_off:self.table_off + len(recs)].copy_(host)
            self._table_ptrs = ptrs

    def _streams(self):
        """Raw handles of the run's streams: the current stream and the helper streams placed for it."""
        from . import streams
        ss = streams.current_set(self.device)
        st = self.streams
        st[0], st[1], st[2], st[3] = ss.main.cuda_stream, ss.side.cuda_stream, ss.trail.cuda_stream, ss.comm.cuda_stream
        self.stream_objs = (ss.main, ss.side, ss.trail, ss.comm)

    def forward(self, video, audio, grad_flat, zero_grad):
        """Issue the forward program; returns (v_emb, a_emb, arena)."""
        dev = self.device
        self._refresh_table()
        fa = torch.empty(self.fa_bytes, dtype=torch.uint8, device=dev)
        self._fill_slots(fa, None, grad_flat, video, audio, None, None)
        self._streams()
        self.fwd_prog[self._zero_index].op = OP_MEMSET0 if (zero_grad and grad_flat is not None) else 0
        lib.call("avid_program_run", self.fwd_prog, 0, self.n_fwd, self.slots, self.n_slots, self.streams, self.ws_arr, 4)
        B = self.vshape[0]
        D = self.vemb.shape[1]
        ve = fa[self.vemb.ref[1]:self.vemb.ref[1] + 4 * B * D].view(torch.float32).view(B, D)
        ae = fa[self.aemb.ref[1]:self.aemb.ref[1] + 4 * B * D].view(torch.float32).view(B, D)
        return ve, ae, fa

    def backward(self, fa, video, audio, dv, da, grad_flat, begin=0, end=None, ba=None):
        dev = self.device
        if ba is None:
            ba = torch.empty(self.ba_bytes, dtype=torch.uint8, device=dev)
        self._fill_slots(fa, ba, grad_flat, video, audio, dv, da)
        self._streams()
        lib.call("avid_program_run", self.bwd_prog, begin, self.n_bwd if end is None else end, self.slots, self.n_slots,
                 self.streams, self.ws_arr, 4)
        return ba


# ------------------------------------------------------------------------------------------------------------------
# autograd node + dispatch
# ------------------------------------------------------------------------------------------------------------------
_ENGINE = None          # the parallel.TrainStep driving the current step (flat gradient buffer, buckets), or None


class engine:
    """``with plan.engine(train_step): ...`` — model calls inside write their parameter gradients into the engine's
    flat gradient buffer (and report them to its gradient buckets) instead of handing tensors to autograd."""

    def __init__(self, eng):
        self.eng = eng

    def __enter__(self):
        global _ENGINE
        self.prev, _ENGINE = _ENGINE, self.eng
        return self

    def __exit__(self, *exc):
        global _ENGINE
        _ENGINE = self.prev
        return False


def _engine_flat(eng, pl):
    """The engine's flat gradient buffer if its layout is the plan's (same parameters, same offsets), else None."""
    flat = eng.flat
    ok = getattr(pl, "_engine_ok", None)
    if ok is None or ok[0] is not flat:
        same = (len(flat.params) == len(pl.params) and all(a is b for a, b in zip(flat.params, pl.params))
                and list(flat.offsets) == list(pl.goff))
        pl._engine_ok = ok = (flat, same)
    return flat.grad if ok[1] else None


class NetFn(torch.autograd.Function):
    """(video, audio) -> (video_emb, audio_emb): models/av_wrapper.py:50-61 with both towers, both heads and all of
    their backward as two launch programs."""

    @staticmethod
    def forward(ctx, video, audio, pl, *params):
        eng = _ENGINE
        gflat = _engine_flat(eng, pl) if eng is not None else None
        ve, ae, fa = pl.forward(video, audio, gflat, zero_grad=gflat is not None)
        ctx.pl, ctx.fa, ctx.video, ctx.audio = pl, fa, video, audio
        ctx.in_engine = gflat is not None
        # the engine of THIS forward pass: the reference's loop calls loss.backward() itself, outside any `with plan.engine`
        # (parallel.DistributedDataParallel), and the gradients still have to land in the flat buffer this forward zeroed
        ctx.eng = eng if gflat is not None else None
        ctx.versions = sum(p._version for p in params)
        ctx.params = params
        return ve, ae

    @staticmethod
    def backward(ctx, dv, da):
        pl, params = ctx.pl, ctx.params
        if ctx.fa is None:
            raise RuntimeError("avid_hip.plan: backward through the same forward pass a second time (its activations "
                               "were released)")
        if sum(p._version for p in params) != ctx.versions:
            raise RuntimeError("avid_hip.plan: a parameter was modified in place between the forward and the backward pass")
        dv, da = dv.contiguous(), da.contiguous()
        eng = ctx.eng
        gflat = _engine_flat(eng, pl) if eng is not None else None
        if gflat is not None:
            eng._plan_backward(pl, ctx.fa, ctx.video, ctx.audio, dv, da)
            ctx.fa = None
            return (None, None, None) + (None,) * len(params)
        g = torch.empty(pl.gnumel, dtype=torch.float32, device=dv.device)
        pl.backward(ctx.fa, ctx.video, ctx.audio, dv, da, g)
        ctx.fa = None
        views = pl.grad_views(g)
        return (None, None, None) + tuple(views[id(p)] for p in params)


def _plan_grad_views(self, g):
    out = {}
    for p, o in zip(self.params, self.goff):
        out[id(p)] = g[o:o + p.numel()].as_strided(p.shape, p.stride())
    return out


Plan.grad_views = _plan_grad_views


def _plan_segments(self, bucket_of, counts):
    """Cut the backward program where a gradient bucket becomes complete: [(end record, [(param index, stream)])] —
    the engine runs the records up to ``end``, reports those gradients (which launches the bucket's all-reduce) and
    goes on.  ``bucket_of[i]`` = bucket of parameter i (the plan's order), ``counts[b]`` = parameters in bucket b."""
    key = (tuple(bucket_of), tuple(counts))
    hit = self.__dict__.setdefault("_segments", {}).get(key)
    if hit is not None:
        return hit
    left = list(counts)
    segs, cur = [], []
    for end, stream, ps in self.grad_ready:
        cut = False
        for i in ps:
            cur.append((i, stream))
            left[bucket_of[i]] -= 1
            cut |= left[bucket_of[i]] == 0
        if cut:
            segs.append((end, cur))
            cur = []
    if cur or not segs or segs[-1][0] != self.n_bwd:
        segs.append((self.n_bwd, cur))
    self._segments[key] = segs
    return segs


Plan.segments = _plan_segments


def eligible(model, video, audio):
    """The launch-program path applies: training step on the GPU through the stock module tree, nothing hooked."""
    if not ENABLED or not (model.training and torch.is_grad_enabled() and video.is_cuda and audio.is_cuda):
        return False
    if video.dtype != torch.float32 or audio.dtype != torch.float32 or video.dim() != 5 or audio.dim() != 4:
        return False
    if torch.cuda.is_current_stream_capturing():
        return False
    mods = model.__dict__.get("_avid_modules")
    if mods is None:
        mods = model.__dict__["_avid_modules"] = list(model.modules())
    for m in mods:
        if m._forward_hooks or m._forward_pre_hooks or m._backward_hooks or m._backward_pre_hooks or not m.training:
            return False
    return True


def run(model, video, audio):
    """``AV_Wrapper.forward`` through the launch programs, or None when the model / call is outside what the compiler
    knows (the caller then takes the per-layer path)."""
    if not eligible(model, video, audio):
        return None
    eng = _ENGINE
    timing = lib.TIMING
    overlap = (not timing) and os.environ.get("AVID_OVERLAP_TOWERS", "1") == "1" and (eng is None or model.overlap_towers)
    trailing = (not timing) and bool(ops.DEFER_WGRAD) and (eng is None or eng._defer_ok())
    video, audio = video.contiguous(), audio.contiguous()
    # (what a compiled Plan froze besides shapes and switches — which parameters take gradients, their dtype, and
    #  the identity of the BatchNorm buffers its slot table points at: parameters frozen after the first step, `.double()`,
    #  `.cpu().cuda()` must not meet a stale program.  ~20 us per call for 141 parameters.)
    mods = model.__dict__.get("_avid_param_list")
    if mods is None:
        mods = model.__dict__["_avid_param_list"] = list(model.parameters())
    fp, dt_ok = 0, True
    for p in mods:
        fp = (fp << 1) | int(p.requires_grad)
        dt_ok = dt_ok and p.dtype == torch.float32 and p.is_cuda
    if not dt_ok:
        return None
    flat = getattr(model, "_bn_flat", None)
    key = (tuple(video.shape), tuple(audio.shape), video.device.index, overlap, trailing, bool(ops.GROUP_WGRAD),
           ops.FUSE_BN_BWD, ops.wino_epoch(), fp, flat.data_ptr() if flat is not None else 0)
    plans = model.__dict__.setdefault("_avid_plans", {})
    pl = plans.get(key)
    if pl is not None and pl is not False and eng is not None and _engine_flat(eng, pl) is None:
        return None                                   # (the engine's gradient buffer is laid out differently)
    if pl is None:
        try:
            pl = Plan(model, video.shape, audio.shape, video.device, overlap, trailing, bool(ops.GROUP_WGRAD))
        except Unsupported:
            pl = False
        plans[key] = pl
    if pl is False:
        return None
    return NetFn.apply(video, audio, pl, *pl.params_fwd)


_OP_NAMES = {0: "nop", 1: "wait", 2: "memset0", 3: "conv_fwd", 4: "conv_dgrad", 5: "conv_wgrad", 6: "wgrad_group", 7: "wgrad_item",
             8: "bn_fwd", 9: "bn_bwd", 10: "bn_pool_fwd", 11: "bn_pool_bwd", 12: "gpool_fwd", 13: "gpool_bwd", 14: "relu_bwd",
             15: "colsum", 16: "wt_batch", 17: "adam"}


def dump(prog, n):
    """Readable listing of a program (debugging / tests)."""
    lines = []
    for k in range(n):
        r = prog[k]
        d = r.d
        geo = (f" [{d.B}x{d.Ti}x{d.Hi}x{d.Wi}x{d.Cin} -> {d.To}x{d.Ho}x{d.Wo}x{d.Cout} k{d.kt}{d.kh}{d.kw} s{d.st}{d.sh}{d.sw}]"
               if r.op in (3, 4, 5, 7) else "")
        refs = " ".join(f"{r.t[j].slot}:{r.t[j].off}" if r.t[j].slot >= 0 else "-" for j in range(NREF))
        lines.append(f"{k:4d} s{r.stream} {_OP_NAMES.get(r.op, r.op):12s}{geo} i={list(r.i)} n={list(r.n)} | {refs}")
    return "\n".join(lines)
