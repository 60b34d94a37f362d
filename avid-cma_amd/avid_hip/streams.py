"""Placement of the step's streams on the hardware's dispatch pipes.

A training step uses four streams: the compute stream (video tower), the audio tower's, one trailing stream for the
weight gradients, and one that carries the gradient collectives.  HIP maps streams onto ``GPU_MAX_HW_QUEUES`` hardware
queues in creation order, and the queues onto FOUR dispatch pipes.  Measured on MI355X (tools/queue_probe2.py,
DESIGN.md §5b):

* two streams on the same hardware queue run in order (expected);
* two streams on different queues of the SAME pipe do not overlap either — the pipe serves one queue at a time — and
  when one of them waits for an event of the other, the pipe sits in the waiting queue's barrier packet until a time
  slice expires: every dispatch of the step then costs ~60 us more (11.3 -> 19-23 ms per step).  Which streams collide
  depends on how many streams the process created before (an RCCL communicator creates its own), i.e. it comes and
  goes with ``GPU_MAX_HW_QUEUES``, with a process group, with two more streams in a test.

So the streams are not taken as created: a pool of candidates is PROBED against the compute stream and against each
other — a chain of single-wave ~25 us kernels on each of two streams; serialised pairs take twice as long as concurrent
ones, there is nothing in between — and the three helper streams are the first candidates that run concurrently with
everything chosen before.  ~10-40 ms once per (device, compute stream).
"""
from __future__ import annotations

import ctypes as C
import os
import time

import torch

from . import lib

_PLACED = {}
PROBE_US = 25
PROBE_ROUNDS = 10
MAX_CANDIDATES = 24


class StreamSet:
    __slots__ = ("main", "side", "trail", "comm", "report")

    def __init__(self, main, side, trail, comm, report):
        self.main, self.side, self.trail, self.comm, self.report = main, side, trail, comm, report


def _pair_us(a, b):
    """Wall time per round of one probe kernel on each of the two streams (issued alternately, waited for together)."""
    ha, hb = C.c_void_p(a.cuda_stream), C.c_void_p(b.cuda_stream)
    a.synchronize()
    b.synchronize()
    t0 = time.perf_counter()
    for _ in range(PROBE_ROUNDS):
        lib.call("avid_probe_spin", PROBE_US, ha)
        lib.call("avid_probe_spin", PROBE_US, hb)
    a.synchronize()
    b.synchronize()
    return (time.perf_counter() - t0) / PROBE_ROUNDS * 1e6


def _alone_us(a):
    ha = C.c_void_p(a.cuda_stream)
    a.synchronize()
    t0 = time.perf_counter()
    for _ in range(PROBE_ROUNDS):
        lib.call("avid_probe_spin", PROBE_US, ha)
    a.synchronize()
    return (time.perf_counter() - t0) / PROBE_ROUNDS * 1e6


def concurrent(a, b, alone_us):
    """True if kernels of the two streams overlap (pair time ~ one kernel), False if the hardware serialises them."""
    return _pair_us(a, b) < 1.5 * alone_us


def place(device=None):
    """The StreamSet for the CURRENT stream of ``device``: helper streams that run concurrently with it and with
    each other.  Cached per (device, current stream)."""
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device())
    main = torch.cuda.current_stream(device)
    key = (device.index, main.cuda_stream)
    hit = _PLACED.get(key)
    if hit is not None:
        return hit
    # as created: inside a graph capture (nothing can be timed; a replayed graph's branches are scheduled by the graph
    # anyway), or with the probe switched off (debugging the placement itself)
    if torch.cuda.is_current_stream_capturing() or os.environ.get("AVID_STREAM_PROBE", "1") != "1":
        s = [torch.cuda.Stream(device) for _ in range(3)]
        hit = _PLACED[key] = StreamSet(main, s[0], s[1], s[2], {"probed": False})
        _register(device, hit)
        return hit
    with torch.cuda.device(device):
        _alone_us(main)                                       # (first launch: module load)
        alone = min(_alone_us(main), _alone_us(main))
        chosen, pool, tried, rejected = [main], [], 0, 0
        while len(chosen) < 4 and tried < MAX_CANDIDATES:
            cand = torch.cuda.Stream(device)
            tried += 1
            pool.append(cand)
            if all(concurrent(c, cand, alone) for c in chosen):
                chosen.append(cand)
            else:
                rejected += 1
        while len(chosen) < 4:                                # fewer than four independent pipes reachable: share
            chosen.append(chosen[-1] if len(chosen) > 1 else torch.cuda.Stream(device))
    del pool                                                  # (rejected candidates are destroyed)
    report = {"probed": True, "alone_us": round(alone, 1), "candidates": tried, "rejected": rejected,
              "independent": len(set(id(c) for c in chosen))}
    hit = _PLACED[key] = StreamSet(main, chosen[1], chosen[2], chosen[3], report)
    _register(device, hit)
    return hit


_BY_MEMBER = {}


def _register(device, ss):
    """Reverse lookup: every helper stream of a set names the set (first registration wins: a helper never becomes the
    compute stream of another set behind the step's back)."""
    for st in (ss.side, ss.trail, ss.comm):
        _BY_MEMBER.setdefault((device.index, st.cuda_stream), ss)


def current_set(device):
    """The StreamSet the current stream belongs to — as its compute stream, or as one of its helpers: autograd runs the
    audio tower's backward nodes with the tower's stream current, the per-layer path reports deferred weight gradients
    with the trailing stream current (``GradBuckets.ready`` -> ``_issue``), a collective may be issued from the
    collectives' stream.  Only a stream that belongs to no set is placed: probing (host-blocking, up to 24 new streams)
    never happens in the middle of a backward pass, and a bucket never goes out on a second comm stream."""
    raw = torch._C._cuda_getCurrentRawStream(device.index)
    hit = _PLACED.get((device.index, raw))
    if hit is None:
        hit = _BY_MEMBER.get((device.index, raw))
    return hit if hit is not None else place(device)


def report(device=None):
    """What the placement of the current stream found (bench.py records it)."""
    return place(device).report
