"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/avid_hip.h declares (no compute calls — there is no GPU here), the ctypes table covers the
header, and the error convention works."""
import ctypes
import os
import re

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(REPO, "include", "avid_hip.h")


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(avid_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported():
    from avid_hip import lib
    syms = declared_symbols()
    assert len(syms) >= 30
    dll = ctypes.CDLL(lib.LIB_PATH)
    for s in syms:
        assert hasattr(dll, s), f"{s} declared in include/avid_hip.h but not exported"
    assert sorted(lib.SIGNATURES) == syms, "ctypes table and header disagree"


def test_conv_desc_matches_header():
    from avid_hip.lib import ConvDesc
    src = open(HEADER).read()
    body = re.search(r"typedef struct avid_conv_desc \{(.*?)\} avid_conv_desc;", src, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = [f.strip() for decl in re.findall(r"int32_t ([^;]+);", body) for f in decl.split(",")]
    assert fields == [n for n, _ in ConvDesc._fields_]
    assert ctypes.sizeof(ConvDesc) == 4 * len(fields)


def test_error_convention():
    from avid_hip import lib
    assert lib.version() >= 100
    d = lib.ConvDesc()          # all zeros -> AVID_E_SHAPE, message set, no kernel launched
    rc = lib.raw("avid_conv_fwd")(ctypes.byref(d), None, None, None, None, None, 0, None, None, None, 0, None)
    assert rc < 0 and "conv" in lib.last_error()
    with pytest.raises(lib.AvidHipError):
        lib.call("avid_l2norm_fwd", 0, 0, None, None, None, None)
    assert lib.raw("avid_bn_workspace_bytes")(1 << 20, 64) > 0


def test_ops_refuse_cpu_tensors():
    import torch
    from avid_hip import ops, AvidHipError
    x = torch.zeros(2, 1, 4, 4, 64)
    w = ops.make_weight(64, 64, 1, 3, 3)
    with pytest.raises(AvidHipError):
        ops.conv_cl(x, w, (1, 1, 1), (0, 1, 1))
    with pytest.raises(AvidHipError):
        ops.l2_normalize(torch.zeros(2, 128))


def test_program_executor_rejects_malformed_programs():
    """avid_program_run validates a program before it launches anything (no GPU needed): null arguments, an unknown record
    kind, a stream index outside the run's table — negative return code, message naming the record."""
    import ctypes as C
    from avid_hip import lib
    prog = (lib.Instr * 2)()
    slots = (C.c_void_p * 4)()
    streams = (C.c_void_p * 2)()
    ws = (lib.StreamWs * 2)()
    run = lib.raw("avid_program_run")
    assert run(None, 0, 1, slots, 4, streams, ws, 2) < 0
    assert run(prog, 0, 1, slots, 4, streams, ws, 0) < 0
    prog[0].op = 99
    assert run(prog, 0, 1, slots, 4, streams, ws, 2) < 0 and "record 0" in lib.last_error()
    prog[0].op, prog[0].stream = 15, 7                     # a column sum on stream 7 of 2
    assert run(prog, 0, 1, slots, 4, streams, ws, 2) < 0 and "stream 7" in lib.last_error()
    prog[0].op = 1                                         # a wait between streams 0 and 5 of 2
    prog[0].i[0], prog[0].i[1] = 0, 5
    assert run(prog, 0, 1, slots, 4, streams, ws, 2) < 0 and "wait" in lib.last_error()
    # a tensor reference to an empty / out-of-range slot is refused before ANY record is dispatched (a null optional
    # operand would run the record with other semantics): record 1 is bad, record 0 (a memset that would fault) never runs
    for j in range(12):
        prog[0].t[j].slot = prog[1].t[j].slot = -1
    prog[0].op, prog[0].stream, prog[0].n[0] = 2, 0, 1 << 40
    prog[0].t[0].slot = 1
    slots[1] = 0xdead0000
    prog[1].op, prog[1].stream = 15, 0
    prog[1].t[0].slot, prog[1].t[1].slot = 1, 9
    assert run(prog, 0, 2, slots, 4, streams, ws, 2) < 0
    assert "record 1" in lib.last_error() and "slot 9" in lib.last_error() and "nothing was launched" in lib.last_error()
    prog[1].t[1].slot = 2                                  # in range, but empty
    assert run(prog, 0, 2, slots, 4, streams, ws, 2) < 0 and "slot 2" in lib.last_error()
    prog[0].op = prog[1].op = 0                            # nop records run to the end
    assert run(prog, 0, 2, slots, 4, streams, ws, 2) == 0
    need = (C.c_size_t * 2)()
    assert lib.raw("avid_program_workspace_bytes")(prog, 0, 2, 2, need) == 0 and list(need) == [0, 0]


def test_weight_transform_rejects_bad_descriptors():
    """avid_weight_transform (one avid_wt_desc by value) checks its descriptor before it launches anything: null
    pointers, an unknown mode, and shapes the split-bf16 fragment order cannot hold (rows % 64, channels % 32)."""
    import ctypes as C
    from avid_hip import lib
    call = lib.raw("avid_weight_transform")
    buf = (C.c_float * 16)()
    p = C.addressof(buf)
    BADARG, UNSUPPORTED = -1, -4           # include/avid_hip.h: AVID_E_BADARG, AVID_E_UNSUPPORTED
    assert call(None, None) == BADARG
    for desc, code in ((lib.WtDesc(0, p, 64, 1, 64, 0), BADARG), (lib.WtDesc(p, 0, 64, 1, 64, 0), BADARG),
                       (lib.WtDesc(p, p, 64, 1, 64, 7), BADARG), (lib.WtDesc(p, p, 0, 1, 64, 0), BADARG),
                       (lib.WtDesc(p, p, 48, 1, 64, 5), UNSUPPORTED), (lib.WtDesc(p, p, 64, 1, 48, 5), UNSUPPORTED),
                       (lib.WtDesc(p, p, 64, 9, 96, 6), UNSUPPORTED)):
        assert call(C.addressof(desc), None) == code, (desc.Cout, desc.Cin, desc.mode)
        assert lib.last_error()
