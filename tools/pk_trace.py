"""Dev tool (needs a libavid_hip.so built with -DAVID_PK_TRACE): per-workgroup time stamps of one igemm_pk_kernel
launch — where a workgroup's time goes between launch, prologue, the k-loops of its tiles and their epilogues."""
import os, sys, ctypes as C
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "avid-cma_amd"))
import numpy as np, torch
from avid_hip import lib, ops

which = sys.argv[1] if len(sys.argv) > 1 else "spt"
B = 64
dev = torch.device("cuda:0")
cfg = {"spt": (64, 64, (1, 3, 3), (0, 1, 1), (8, 28, 28)), "tmp": (64, 64, (3, 1, 1), (1, 0, 0), (8, 28, 28)),
       "c3": (128, 128, (1, 3, 3), (0, 1, 1), (4, 14, 14))}[which]
cin, cout, k, pd, (T, H, W) = cfg
x = torch.randn(B, T, H, W, cin, device=dev)
if os.environ.get("CB_DATA") == "zero": x.zero_()
w = ops.make_weight(cout, cin, *k).normal_().to(dev)
for _ in range(5): y = ops.conv_cl(x, w, (1, 1, 1), pd)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); y = ops.conv_cl(x, w, (1, 1, 1), pd); e1.record(); torch.cuda.synchronize()
print(f"launch (+ reduce) by events: {e0.elapsed_time(e1)*1e3:.1f} us")
dll = C.CDLL(lib.LIB_PATH)
buf = np.zeros(1024 * 64, dtype=np.int64)
assert dll.avid_debug_pk_trace(buf.ctypes.data_as(C.c_void_p)) == 0
tr = buf.reshape(1024, 64)
G = int((tr[:, 0] != 0).sum())
wall = tr[:G, :32].astype(np.float64) * 0.01   # us (100 MHz)
clk = tr[:G, 32:].astype(np.float64)
t0 = wall[:, 0].min()
print(f"workgroups {G}; kernel span (first start .. last end) {wall[:, 31].max() - t0:.1f} us")
print(f"start skew: mean {np.mean(wall[:,0]-t0):.2f} max {np.max(wall[:,0]-t0):.2f} us;  prologue {np.mean(wall[:,1]-wall[:,0]):.2f} us (max {np.max(wall[:,1]-wall[:,0]):.2f})")
mhz = (clk[:, 31] - clk[:, 0]) / (wall[:, 31] - wall[:, 0])
print(f"shader clock over the kernel: {mhz.mean():.0f} MHz (min {mhz.min():.0f} max {mhz.max():.0f})")
nseg = np.zeros(G, dtype=int)
for j in range(9):
    has = wall[:, 2 + 3 * j] > 0
    if not has.any(): break
    nseg[has] = j + 1
    kl = wall[has, 3 + 3 * j] - wall[has, 2 + 3 * j]
    ep = wall[has, 4 + 3 * j] - wall[has, 3 + 3 * j]
    gap = (wall[has, 2 + 3 * j] - (wall[has, 1] if j == 0 else wall[has, 4 + 3 * (j - 1)]))
    print(f"seg {j}: {int(has.sum()):4d} wgs  k-loop {kl.mean():7.2f} us (min {kl.min():.2f} max {kl.max():.2f})  epilogue {ep.mean():5.2f} (max {ep.max():.2f})  zero/gap {gap.mean():5.2f}  ends at {np.mean(wall[has, 4+3*j]-t0):7.1f} (max {np.max(wall[has, 4+3*j]-t0):.1f})")
end = wall[:, 31] - t0
print(f"workgroup end: mean {end.mean():.1f} min {end.min():.1f} max {end.max():.1f} us; segments per wg: {np.bincount(nseg)}")
# per XCD (blockIdx & 7) and per CU (the two workgroups of a CU are blockIdx b and b + G/2 when G = 2 x CUs)
xcd = np.arange(G) & 7
for xc in range(8):
    m = xcd == xc
    print(f"xcd {xc}: start {np.mean(wall[m,0]-t0):5.2f}  end mean {end[m].mean():6.1f} max {end[m].max():6.1f}  clock {mhz[m].mean():.0f} MHz"
          f"  k-loop/tile {np.mean(wall[m,3]-wall[m,2]):.2f} {np.mean(wall[m,9]-wall[m,8]):.2f}")
if G == 512:
    cu_end = np.maximum(end[:256], end[256:])
    print(f"per-CU end (later of its two workgroups): mean {cu_end.mean():.1f} min {cu_end.min():.1f} max {cu_end.max():.1f}; "
          f"pair difference mean {np.mean(np.abs(end[:256]-end[256:])):.1f} us")
    print("histogram of CU end times:", np.histogram(cu_end, bins=8)[0], np.round(np.histogram(cu_end, bins=8)[1], 0))
