"""Dev tool (needs a library built with -DAVID_PK_TRACE: tools/build_variant.sh trace conv -DAVID_PK_TRACE, then
AVID_HIP_LIB=.../libavid_hip_trace.so): wave 0's time stamps inside one tconv64_kernel launch — per tile: staging (LDS
stores + load issue), products of channel block 0, barrier, products of block 1, epilogue, barrier."""
import os, sys, ctypes as C
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "avid-cma_amd"))
import numpy as np, torch
from avid_hip import lib, ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda:0")
x = torch.randn(B, 8, 28, 28, 64, device=dev)
w = ops.make_weight(64, 64, 3, 1, 1).normal_().to(dev)
for _ in range(5): y = ops.conv_cl(x, w, (1, 1, 1), (1, 0, 0))
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); y = ops.conv_cl(x, w, (1, 1, 1), (1, 0, 0)); e1.record(); torch.cuda.synchronize()
print(f"launch by events: {e0.elapsed_time(e1)*1e3:.1f} us")
dll = C.CDLL(lib.LIB_PATH)
buf = np.zeros(1024 * 64, dtype=np.int64)
assert dll.avid_debug_pk_trace(buf.ctypes.data_as(C.c_void_p)) == 0
tr = buf.reshape(1024, 64)
G = int((tr[:, 0] != 0).sum())
wall = tr[:G, :32].astype(np.float64) * 0.01
clk = tr[:G, 32:].astype(np.float64)
t0 = wall[:, 0].min()
print(f"workgroups {G}; span {wall[:, 31].max() - t0:.1f} us; start skew max {np.max(wall[:,0]-t0):.2f}; prologue {np.mean(wall[:,1]-wall[:,0]):.2f} us")
mhz = (clk[:, 31] - clk[:, 0]) / (wall[:, 31] - wall[:, 0])
print(f"shader clock: {mhz.mean():.0f} MHz")
names = ["zero", "stage0", "products0", "barrier", "stage1+products1", "epilogue", "barrier"]
for t in range(4):
    b = 2 + 7 * t
    if not (wall[:, b] > 0).any(): break
    has = wall[:, b + 6] > 0
    prev = wall[has, 1] if t == 0 else wall[has, b - 1]
    seg = [wall[has, b] - prev] + [wall[has, b + i + 1] - wall[has, b + i] for i in range(6)]
    cyc = [clk[has, b + i + 1] - clk[has, b + i] for i in range(6)]
    print(f"tile {t} ({int(has.sum())} wgs): " + "  ".join(f"{n} {v.mean():.2f}" for n, v in zip(["gap"] + names[1:], seg)) +
          f"  | total {np.mean(wall[has, b + 6] - wall[has, b]):.2f} us; products0 {cyc[1].mean():.0f} cycles")
# tile 3, channel block 0: every wave's own start / end of its products relative to wave 0's stamp 3 + 7 * 3
ref = wall[:, 24:25]
st = tr[512:512 + G, 0:8].astype(np.float64) * 0.01 - ref
en = tr[512:512 + G, 8:16].astype(np.float64) * 0.01 - ref
print("tile 3 products0 per wave: start " + " ".join(f"{v:.2f}" for v in st.mean(0)) + " | end " + " ".join(f"{v:.2f}" for v in en.mean(0)) +
      f" | last wave ends {en.max(1).mean():.2f}, barrier leaves at {np.mean(wall[:, 26] - wall[:, 24]):.2f}")
for wg in (0, 100):
    print("raw wg", wg, " ".join(f"{v - wall[wg, 0]:.2f}" for v in wall[wg, :32]))
    print("   per-wave start", " ".join(f"{v * 0.01 - wall[wg, 0]:.2f}" for v in tr[512 + wg, 0:8]), "end", " ".join(f"{v * 0.01 - wall[wg, 0]:.2f}" for v in tr[512 + wg, 8:16]))
end = wall[:, 31] - t0
print(f"end: mean {end.mean():.1f} min {end.min():.1f} max {end.max():.1f}")
