"""The development tools (tools/*.py, *.sh, *.hip) are not product code, but DESIGN.md's numbers rest on them: a tool that a
kernel change has broken should be noticed when it breaks, not when its number is next needed.  No GPU here:
every Python tool must compile, every C-ABI entry point a tool names must still exist with the signature table's arity, every
environment switch it sets must still be read somewhere in the package, shell scripts must parse, and the lab kernels must still
compile for gfx950 (hipcc cross-compiles)."""
import ast
import glob
import os
import re
import subprocess

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOLS = os.path.join(REPO, "tools")
PY = sorted(glob.glob(os.path.join(TOOLS, "*.py")))
SH = sorted(glob.glob(os.path.join(TOOLS, "*.sh")))


@pytest.mark.parametrize("path", PY, ids=[os.path.basename(p) for p in PY])
def test_python_tools_compile_and_name_live_entry_points(path):
    from avid_hip import lib
    src = open(path).read()
    tree = ast.parse(src, path)                                    # syntax
    # lib.raw("avid_x") / lib.call("avid_x", ...) / dll.avid_x(...): the symbol must be exported (ctypes.CDLL resolves lazily)
    names = set(re.findall(r'lib\.(?:raw|call)\(\s*"(avid_\w+)"', src)) | set(re.findall(r"dll\.(avid_\w+)\(", src))
    for n in sorted(names):
        if n.startswith(("avid_debug_pk_trace", "avid_debug_wino_trace", "avid_debug_ww_trace")):
            continue                                                # only in -DAVID_PK_TRACE / -DAVID_WINO_TRACE / -DAVID_WW_TRACE builds (tools/build_variant.sh)
        assert hasattr(lib._lib, n), f"{os.path.basename(path)} calls {n}, which libavid_hip.so no longer exports"
    # attributes of the package's modules the tool reaches for
    import avid_hip.ops as ops
    import avid_hip.parallel as parallel
    for mod, obj in (("ops", ops), ("parallel", parallel), ("lib", lib)):
        for attr in set(re.findall(r"(?<![\w.])%s\.([A-Za-z_]\w*)" % mod, src)):
            imported = any(isinstance(nd, (ast.Import, ast.ImportFrom)) and any(a.name.split(".")[-1] == mod or a.asname == mod for a in nd.names)
                           for nd in ast.walk(tree))
            if imported:
                assert hasattr(obj, attr), f"{os.path.basename(path)} uses {mod}.{attr}, which no longer exists"


def test_environment_switches_the_tools_set_are_still_read():
    read = ""
    for pat in ("avid-cma_amd/csrc/*.hip", "avid-cma_amd/csrc/*.h", "avid-cma_amd/avid_hip/*.py", "avid-cma_amd/models/*.py",
                "avid-cma_amd/criterions/*.py", "bench.py", "tools/*.py", "tools/*.sh", "tests/*.py"):
        for f in glob.glob(os.path.join(REPO, pat)):
            read += open(f).read()
    readers = set(re.findall(r'getenv\("(AVID_\w+)"\)', read)) | set(re.findall(r"""environ(?:\.get|\.setdefault)?\(?\[?\s*["'](AVID_\w+)["']""", read))
    readers |= set(re.findall(r"#\s*ifn?def\s+(AVID_\w+)", read)) | set(re.findall(r"#\s*if\s+(AVID_\w+)", read))
    for path in PY + SH:
        src = open(path).read()
        for sw in set(re.findall(r"\b(AVID_[A-Z0-9_]+)=", src)):
            assert sw in readers, f"{os.path.basename(path)} sets {sw}, which nothing reads any more"


@pytest.mark.parametrize("path", SH, ids=[os.path.basename(p) for p in SH])
def test_shell_tools_parse(path):
    assert subprocess.run(["bash", "-n", path], capture_output=True).returncode == 0


@pytest.mark.parametrize("name", ["split_dot_check", "mfma_rate", "split_lab"])
def test_lab_kernels_still_compile(name, tmp_path):
    """hipcc cross-compiles the lab kernels DESIGN.md 8e-8g quote (device code only: a few seconds each)."""
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    src = os.path.join(TOOLS, name + ".hip")
    out = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-c", "--cuda-device-only", src, "-o", str(tmp_path / "o.o")],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-800:]


# ---- register / scratch budget of the kernels the bs-64 step launches: read from the code objects'
# metadata notes of the built library (tools/kernel_resources.py).  STEP_KERNELS are the instantiations rocprofv3 sees in a
# default bench.py run (profiles/r0*_kernel_stats.csv); ALLOWED lists the ones that still carry a private segment, with the
# bytes they are allowed — a number here may only go down.
STEP_KERNEL_PREFIXES = ("igemm_pk_kernel<4, 1, 1, 2, 0, false, 0, true>", "igemm_pk_kernel<4, 1, 1, 2, 0, false, 1, true>",
                        "igemm_pk_kernel<4, 1, 1, 2, 0, false, 2, false>", "igemm_pk_kernel<4, 1, 1, 2, 0, false, 3, false>",
                        "igemm_pk_kernel<4, 1, 1, 2, 1, false, 0, true>", "igemm_pk_kernel<4, 1, 1, 2, 1, false, 8, true>",
                        "igemm_pk_kernel<4, 1, 1, 2, 1, false, 9, true>", "igemm_pk_kernel<4, 1, 1, 2, 1, true, 8, true>",
                        "igemm_pk_kernel<4, 1, 1, 2, 1, true, 9, true>", "igemm_pk_kernel<4, 1, 1, 4, 0, false, 0, true>",
                        "igemm_pk_kernel<4, 1, 1, 4, 0, false, 1, true>", "igemm_pk_kernel<4, 1, 1, 4, 1, false, 8, true>",
                        "wino2p_kernel<1>", "wino2p_kernel<2>", "wino2p_kernel<4>", "wino2p_kernel<6>",
                        "wino_kernel<1>", "wino_kernel<4>", "wino_kernel<6>", "wino_wgrad_kernel", "tconv64_kernel<0, 0, true>",
                        "tconv64_kernel<0, 1, true>", "tconv64_kernel<0, 0, false>", "tconv64_kernel<0, 1, false>",
                        "twgrad64_kernel<true>", "twgrad64_kernel<false>", "stem_fwd3p_kernel<3, 3, 1>", "stem_wgrad3_kernel<3, 3, true>",
                        "wgrad_group_kernel<true, true>", "wgrad_tab_kernel<1, 3, true>", "xmodal_fused_kernel<false, 64, false>",
                        "adam_flat_kernel", "bn_apply_kernel", "bn_bwd_apply_kernel", "bn_fin_apply_kernel", "bn_bwd_fin_apply_kernel")
SCRATCH_ALLOWED = {}        # name: (vgpr spills, private-segment bytes) of a step kernel that is allowed any: none since round 6


def test_step_kernels_have_no_scratch():
    from tools_path import kernel_table
    rows = {r["name"]: r for r in kernel_table()}
    missing = [k for k in STEP_KERNEL_PREFIXES if k not in rows]
    assert not missing, ("instantiations the step is supposed to launch are not in libavid_hip.so", missing)
    bad = {}
    for k in STEP_KERNEL_PREFIXES:
        r = rows[k]
        spills, scratch = SCRATCH_ALLOWED.get(k, (0, 0))
        if r["vgpr_spill_count"] > spills or r["private_segment_fixed_size"] > scratch:
            bad[k] = (r["vgpr_spill_count"], r["private_segment_fixed_size"])
    assert not bad, f"(vgpr spills, scratch bytes) of step kernels: {bad}"
