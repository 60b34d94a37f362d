"""Register / scratch budget of every kernel in libavid_hip.so, read from the code objects' metadata notes (no GPU):
    python tools/kernel_resources.py [--spills] [pattern ...]
prints  vgpr  agpr  sgpr  vgpr_spills  sgpr_spills  scratch_bytes  lds_bytes  kernel  for every kernel whose demangled name
contains one of the patterns (all kernels without one); --spills keeps only kernels with spilled registers or a private
segment.  tests/test_tools.py imports `kernel_table()` and holds the instantiations the bs-64 step launches to their budgets.

How: `llvm-objdump --offloading` unbundles the gfx950 code objects of the shared library (one per translation unit),
`llvm-readelf --notes` prints their amdhsa.kernels metadata."""
import glob
import os
import re
import shutil
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(REPO, "avid-cma_amd", "avid_hip", "libavid_hip.so")
LLVM = "/opt/rocm/lib/llvm/bin"
FIELDS = ("vgpr_count", "agpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size",
          "group_segment_fixed_size", "uses_dynamic_stack")


def kernel_table(lib=LIB):
    """[{name (demangled, without the `avid::` namespace and the argument list), vgpr_count, ..., private_segment_fixed_size}]"""
    tmp = tempfile.mkdtemp(prefix="avid_co_")
    try:
        so = os.path.join(tmp, "lib.so")
        shutil.copy(lib, so)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", so], check=True, capture_output=True, cwd=tmp)
        import yaml
        rows = []
        for co in sorted(glob.glob(os.path.join(tmp, "lib.so.*gfx950*"))):
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], check=True, capture_output=True, text=True).stdout
            for doc in re.findall(r"^\s*---\n(.*?)^\s*\.\.\.", notes, flags=re.S | re.M):
                meta = yaml.safe_load(doc)
                for k in (meta or {}).get("amdhsa.kernels", []):
                    r = {f: int(k.get("." + f, 0) or 0) for f in FIELDS}
                    r["mangled"] = k[".name"]
                    rows.append(r)
        filt = shutil.which("llvm-cxxfilt") or shutil.which("c++filt")
        dem = subprocess.run([filt], input="\n".join(r["mangled"] for r in rows), capture_output=True, text=True,
                             check=True).stdout.splitlines() if filt else [r["mangled"] for r in rows]
        for r, d in zip(rows, dem):
            d = re.sub(r"^void ", "", d)
            depth, cut = 0, len(d)
            for i, ch in enumerate(d):                 # the argument list: the first '(' outside template brackets
                depth += ch == "<"
                depth -= ch == ">"
                if ch == "(" and depth == 0:
                    cut = i
                    break
            r["name"] = d[:cut].replace("avid::", "", 1)
        return rows
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    only_spills = "--spills" in sys.argv
    print(f"{'vgpr':>5s} {'agpr':>5s} {'sgpr':>5s} {'vspill':>6s} {'sspill':>6s} {'scratch':>7s} {'lds':>7s}  kernel")
    for r in sorted(kernel_table(), key=lambda r: r["name"]):
        if args and not any(a in r["name"] for a in args):
            continue
        if only_spills and not (r.get("vgpr_spill_count") or r.get("private_segment_fixed_size")):
            continue
        print(f"{r.get('vgpr_count', 0):5d} {r.get('agpr_count', 0):5d} {r.get('sgpr_count', 0):5d} {r.get('vgpr_spill_count', 0):6d} "
              f"{r.get('sgpr_spill_count', 0):6d} {r.get('private_segment_fixed_size', 0):7d} {r.get('group_segment_fixed_size', 0):7d}  {r['name']}")


if __name__ == "__main__":
    main()
