"""Dev tool (libavid_hip.so built with -DAVID_WINO_TRACE): where a workgroup of wino_kernel spends its time — summed per
phase over its units: 0 loop top, 1 input transform (+ barriers), 2 products, 3 (chunk loop exit), 4 output transform."""
import os, sys, ctypes as C
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "avid-cma_amd"))
import numpy as np, torch
from avid_hip import lib, ops
dev = torch.device("cuda:0")
cin = cout = int(sys.argv[1]) if len(sys.argv) > 1 else 64
T, H, W = (8, 28, 28) if cin == 64 else (4, 14, 14)
x = torch.randn(64, T, H, W, cin, device=dev)
w = ops.make_weight(cout, cin, 1, 3, 3).normal_().to(dev)
for _ in range(3): y = ops.conv_cl(x, w, (1, 1, 1), (0, 1, 1))
torch.cuda.synchronize()
dll = C.CDLL(os.path.join(REPO, "avid-cma_amd", "avid_hip", "libavid_hip.so"))
buf = np.zeros(1024 * 8, dtype=np.int64)
assert dll.avid_debug_wino_trace(buf.ctypes.data_as(C.c_void_p)) == 0
tr = buf.reshape(1024, 8)[:512].astype(np.float64) * 0.01
names = ["loop top", "input transform", "products", "chunk-loop exit", "output transform + stores"]
tot = tr[:, :5].sum(1)
print(f"per workgroup total {tot.mean():.1f} us (min {tot.min():.1f} max {tot.max():.1f})")
for i, n in enumerate(names):
    print(f"  {n:28s} {tr[:, i].mean():7.2f} us  ({100 * tr[:, i].mean() / tot.mean():4.1f} %)")
