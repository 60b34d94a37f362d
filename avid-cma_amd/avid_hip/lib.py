"""ctypes binding of libavid_hip.so (the C-ABI declared in include/avid_hip.h).

The HIP library is THE compute path: there is no CPU / eager fallback.  If the shared object is
missing the import of this module raises, loudly (``AVID_HIP_AUTOBUILD=1`` lets it try ``make``
once first, which is what ``__graft_entry__.build()`` does explicitly).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

# HIP multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (default 4).  An RCCL communicator takes
# queues of its own; with the default, the audio tower's side stream then shares a queue with the main stream
# and the towers serialise (measured: +0.5..0.9 ms per step the moment a process group exists, collectives or
# not; 8 queues: no difference to the single-process step).  Read by the HIP runtime when it initialises, so
# this has to happen before the first HIP call of the process.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

# torch must be imported BEFORE libavid_hip.so is dlopen'ed: torch ships its own libamdhip64.so, and a process
# that loads /opt/rocm's copy first (through this library's DT_NEEDED) ends up with two HIP runtimes — the
# one this library is bound to then reports "no ROCm-capable device" on its first launch.
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
_PKG = os.path.dirname(_HERE)
# AVID_HIP_LIB: a development switch (tools/build_variant.sh) — A/B of two builds of the library on one GPU box
LIB_PATH = os.environ.get("AVID_HIP_LIB") or os.path.join(_HERE, "libavid_hip.so")

AVID_OK = 0


class BnBwdFuse(C.Structure):
    """Mirror of ``avid_bn_bwd_fuse`` (include/avid_hip.h)."""
    _fields_ = [("x", C.c_void_p), ("scale", C.c_void_p), ("shift", C.c_void_p), ("mean", C.c_void_p),
                ("invstd", C.c_void_p), ("relu", C.c_int32), ("partials", C.c_void_p)]


class InAffine(C.Structure):
    """Mirror of ``avid_in_affine`` (include/avid_hip.h): the BatchNorm (+ReLU) a convolution applies to its input."""
    _fields_ = [("scale", C.c_void_p), ("shift", C.c_void_p), ("relu", C.c_int32)]


class ConvDesc(C.Structure):
    """Mirror of ``avid_conv_desc`` (include/avid_hip.h)."""
    _fields_ = [(n, C.c_int32) for n in (
        "B", "Ti", "Hi", "Wi", "Cin", "To", "Ho", "Wo", "Cout", "kt", "kh", "kw",
        "st", "sh", "sw", "pt", "ph", "pw", "x_channel_first")]


class WgradItem(C.Structure):
    """Mirror of ``avid_wgrad_item`` (include/avid_hip.h)."""
    _fields_ = [("d", ConvDesc), ("x", C.c_void_p), ("dy", C.c_void_p), ("dw", C.c_void_p)]


class WtDesc(C.Structure):
    """avid_wt_desc (include/avid_hip.h)."""
    _fields_ = [("w", C.c_void_p), ("wt", C.c_void_p), ("Cout", C.c_int32), ("ntaps", C.c_int32), ("Cin", C.c_int32),
                ("mode", C.c_int32)]


class Ref(C.Structure):
    """Mirror of ``avid_ref``: (slot, byte offset); slot < 0 = NULL."""
    _fields_ = [("slot", C.c_int32), ("reserved", C.c_int32), ("off", C.c_int64)]


INSTR_REFS = 12


class Instr(C.Structure):
    """Mirror of ``avid_instr`` (include/avid_hip.h, "Launch programs")."""
    _fields_ = [("op", C.c_int32), ("stream", C.c_int32), ("mark", C.c_int32), ("reserved", C.c_int32),
                ("d", ConvDesc), ("i", C.c_int32 * 6), ("n", C.c_int64 * 2), ("f", C.c_float * 6), ("t", Ref * INSTR_REFS)]


class StreamWs(C.Structure):
    """Mirror of ``avid_stream_ws``."""
    _fields_ = [("ptr", C.c_void_p), ("bytes", C.c_size_t)]


def build_library(verbose=False):
    """Compile csrc/*.hip for gfx950 into avid_hip/libavid_hip.so (hipcc cross-compiles without a GPU)."""
    proc = subprocess.run(["make", "-C", _PKG, "-j8"], capture_output=True, text=True)
    if verbose or proc.returncode != 0:
        print(proc.stdout[-4000:])
        print(proc.stderr[-4000:])
    if proc.returncode != 0:
        raise RuntimeError("building libavid_hip.so failed (see output above)")
    return LIB_PATH


def _load():
    if not os.path.exists(LIB_PATH):
        if os.environ.get("AVID_HIP_AUTOBUILD", "0") == "1":
            build_library()
        else:
            raise RuntimeError(
                f"{LIB_PATH} is missing: the HIP extension is required (no CPU fallback). "
                "Run `python -c 'import __graft_entry__ as g; g.build()'` or `make -C avid-cma_amd`.")
    return C.CDLL(LIB_PATH)


_lib = _load()

_vp, _i, _i64, _u64, _f, _sz = C.c_void_p, C.c_int, C.c_int64, C.c_uint64, C.c_float, C.c_size_t
_dp = C.POINTER(ConvDesc)

# name -> (restype, argtypes); must list every symbol of include/avid_hip.h (checked by tests)
SIGNATURES = {
    "avid_last_error": (C.c_char_p, []),
    "avid_version": (_i, []),
    "avid_device_info": (_i, [_i, C.POINTER(_i), C.POINTER(_i), C.c_char_p, _i]),
    "avid_timing_enable": (_i, [_i]),
    "avid_timing_report": (_i, [C.c_char_p, _sz]),
    "avid_conv_fwd_workspace_bytes": (_sz, [_dp]),
    "avid_conv_fwd_stats_rows": (_i, [_dp]),
    "avid_conv_fwd": (_i, [_dp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _sz, _vp]),
    "avid_conv_takes_in_affine": (_i, [_dp]),
    "avid_conv_fwd_in": (_i, [_dp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _sz, _vp]),
    "avid_debug_in_affine_launches": (C.c_longlong, [_i]),
    "avid_conv_dgrad_workspace_bytes": (_sz, [_dp]),
    "avid_conv_dgrad_bn_rows": (_i, [_dp]),
    "avid_conv_dgrad": (_i, [_dp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "avid_weight_transpose_batched": (_i, [_i, _vp, _i64, _vp]),
    "avid_weight_transform": (_i, [_vp, _vp]),
    "avid_conv_wgrad_workspace_bytes": (_sz, [_dp]),
    "avid_conv_wgrad": (_i, [_dp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "avid_conv_wgrad_in": (_i, [_dp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "avid_conv_wgrad_groupable": (_i, [_dp]),
    "avid_conv_wgrad_group_workspace_bytes": (_sz, [_i, C.POINTER(WgradItem)]),
    "avid_conv_wgrad_group": (_i, [_i, C.POINTER(WgradItem), _vp, _sz, _vp]),
    "avid_conv_kernel_name": (_i, [_dp, _i, C.c_char_p, _i]),
    "avid_conv_uses_wino": (_i, [_dp, _i]),
    "avid_conv_uses_split": (_i, [_dp, _i]),
    "avid_conv_split_bytes": (_sz, [_dp]),
    "avid_debug_presplit_launches": (C.c_longlong, []),
    "avid_wino_configure": (_i, [_i, _i64, _i]),
    "avid_wino2_configure": (_i, [_i]),
    "avid_wino2_pre_configure": (_i, [_i]),
    "avid_wgrad_pre_configure": (_i, [_i]),
    "avid_stem_fwd_pre_configure": (_i, [_i]),
    "avid_tconv_configure": (_i, [_i]),
    "avid_set_cu_budget": (_i, [_i]),
    "avid_cu_budget": (_i, []),
    "avid_bn_workspace_bytes": (_sz, [_i64, _i]),
    "avid_bn_fwd_train": (_i, [_i64, _i, _vp, _vp, _vp, _vp, _vp, _f, _f, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _sz, _vp]),
    "avid_bn_fwd_eval": (_i, [_i64, _i, _vp, _vp, _vp, _vp, _vp, _f, _i, _vp, _vp, _vp]),
    "avid_bn_bwd": (_i, [_i64, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _vp, _sz, _vp]),
    "avid_bn_relu_maxpool_fwd": (_i, [_i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                      _vp, _i, _vp, _sz, _vp]),
    "avid_bn_relu_maxpool_bwd": (_i, [_i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz,
                                      _vp]),
    "avid_clip_normalize": (_i, [_i, _i, _i, _i, _vp, C.POINTER(_f), C.POINTER(_f), _vp, _vp]),
    "avid_logspec_basis_floats": (_sz, [_i]),
    "avid_logspec_basis": (_i, [_i, _vp, _vp]),
    "avid_logspec_workspace_bytes": (_sz, [_i, _i, _i]),
    "avid_logspec": (_i, [_i, _i, _vp, _i, _i, _i, _vp, _vp, _vp, _f, _vp, _vp, _sz, _vp]),
    "avid_maxpool_hw3s2_fwd": (_i, [_i, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "avid_maxpool_hw3s2_bwd": (_i, [_i, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "avid_global_maxpool_fwd": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp]),
    "avid_global_maxpool_bwd": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp]),
    "avid_relu_bwd": (_i, [_i64, _vp, _vp, _vp, _vp]),
    "avid_colsum": (_i, [_i64, _i, _vp, _vp, _vp]),
    "avid_l2norm_fwd": (_i, [_i, _i, _vp, _vp, _vp, _vp]),
    "avid_l2norm_bwd": (_i, [_i, _i, _vp, _vp, _vp, _vp, _vp]),
    "avid_alias_draw": (_i, [_i64, _i64, _vp, _vp, _i, _u64, _u64, _vp, _vp, _i64, _vp, _vp]),
    "avid_counter_add": (_i, [_vp, _u64, _vp]),
    "avid_bank_scores_fwd": (_i, [_i, _i, _i, _i64, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp]),
    "avid_bank_scores_bwd": (_i, [_i, _i, _i, _i64, _vp, _vp, _vp, _vp, _f, _i, _vp, _vp]),
    "avid_mean_exp": (_i, [_i, _i, _i, _vp, _vp, _vp]),
    "avid_nce_workspace_bytes": (_sz, []),
    "avid_nce_fwd": (_i, [_i, _i, _i, _vp, _i, _vp, _i, _vp, _f, _i, _vp, _vp, _sz, _vp]),
    "avid_nce_bwd": (_i, [_i, _i, _i, _vp, _i, _vp, _i, _vp, _vp, _f, _vp, _vp, _vp]),
    "avid_bank_update": (_i, [_i, _i, _i64, _vp, _vp, _vp, _f, _vp, _vp]),
    "avid_xmodal_fused_workspace_bytes": (_sz, [_i, _i]),
    "avid_xmodal_fused": (_i, [_i, _i, _i, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _f, _vp, _f, _vp, _vp, _vp, _vp, _vp, _vp,
                               _sz, _vp, _vp]),
    "avid_cma_fused_workspace_bytes": (_sz, [_i, _i, _i]),
    "avid_cma_fused": (_i, [_i, _i, _i, _i, _i, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _vp, _f, _f, _vp, _vp, _vp, _vp,
                            _vp, _vp, _sz, _vp, _vp]),
    "avid_bank_update2": (_i, [_i, _i, _i64, _vp, _vp, _vp, _vp, _vp, _f, _f, _vp, _vp]),
    "avid_cma_negatives": (_i, [_i, _i, _i, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "avid_cma_topk_workspace_bytes": (_sz, [_i64, _i, _i]),
    "avid_cma_topk": (_i, [_i64, _i, _vp, _vp, _i64, _i, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "avid_adam_flat": (_i, [_i64, _vp, _vp, _vp, _vp, _f, _f, _f, _f, _f, _i64, _vp, _vp, _f, _vp]),
    "avid_probe_spin": (_i, [_i, _vp]),
    "avid_stream_wait": (_i, [_vp, _vp]),
    "avid_clock_probe": (_i, [_i, _vp, _vp]),
    "avid_program_instr_bytes": (_sz, []),
    "avid_program_workspace_bytes": (_i, [C.POINTER(Instr), _i, _i, _i, C.POINTER(_sz)]),
    "avid_program_run": (_i, [C.POINTER(Instr), _i, _i, C.POINTER(_vp), _i, C.POINTER(_vp), C.POINTER(StreamWs), _i]),
}

for _name, (_res, _args) in SIGNATURES.items():
    _fn = getattr(_lib, _name)          # AttributeError here == a declared symbol is not exported
    _fn.restype = _res
    _fn.argtypes = _args


class AvidHipError(RuntimeError):
    pass


if _lib.avid_program_instr_bytes() != C.sizeof(Instr):
    raise AvidHipError(f"avid_instr is {_lib.avid_program_instr_bytes()} bytes in the library, {C.sizeof(Instr)} in "
                       "avid_hip/lib.py: the ctypes mirror is out of date")


def last_error() -> str:
    msg = _lib.avid_last_error()
    return msg.decode("utf-8", "replace") if msg else ""


def check(rc: int, what: str):
    """Error convention of the C-ABI: negative return -> Python exception (never a silent fallback)."""
    if rc != AVID_OK:
        raise AvidHipError(f"{what} failed (rc={rc}): {last_error()}")


def call(name: str, *args):
    check(getattr(_lib, name)(*args), name)


def raw(name: str):
    return getattr(_lib, name)


def version() -> int:
    return _lib.avid_version()


def device_info(device: int = 0):
    cu, lds = C.c_int(0), C.c_int(0)
    buf = C.create_string_buffer(64)
    check(_lib.avid_device_info(device, C.byref(cu), C.byref(lds), buf, 64), "avid_device_info")
    return {"cu_count": cu.value, "lds_bytes": lds.value, "arch": buf.value.decode()}


TIMING = False     # HIP-event timers on: one event pair per launch — the callers keep the step on ONE stream then


def timing_enable(on: bool):
    global TIMING
    check(_lib.avid_timing_enable(1 if on else 0), "avid_timing_enable")
    TIMING = bool(on)


def timing_report():
    """{kernel name: {"launches", "ms", "flops", "bytes"}} for everything launched since timing_enable(True)."""
    buf = C.create_string_buffer(1 << 16)
    check(_lib.avid_timing_report(buf, len(buf)), "avid_timing_report")
    out = {}
    for line in buf.value.decode().splitlines():
        name, n, ms, fl, by = line.split(";")
        out[name] = {"launches": int(n), "ms": float(ms), "flops": float(fl), "bytes": float(by)}
    return out
