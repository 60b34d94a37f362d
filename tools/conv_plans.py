"""Dev tool (CPU): the kernel + work split the planner picks for every conv shape of the step at a given batch."""
import sys, os, re, ctypes as C
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "avid-cma_amd"))
from avid_hip import lib, ops
src = open(os.path.join(REPO, "tools", "conv_bench.py")).read()
L = eval(re.search(r"L = (\[.*?\n\])", src, re.S).group(1))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
for name, cin, cout, k, st, pd, (T, H, W) in L:
    d = ops._desc((B, T, H, W), cin, cout, k, st, pd, False)
    out = []
    for which in (0, 1):
        buf = C.create_string_buffer(256)
        lib.raw("avid_conv_kernel_name")(C.byref(d), which, buf, 256)
        out.append(buf.value.decode())
    print(f"{name:10s} | {out[0]:58s} | {out[1]}")
