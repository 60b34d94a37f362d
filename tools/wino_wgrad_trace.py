"""Dev tool (library built with -DAVID_WW_TRACE: tools/build_variant.sh wwtrace wino "-DAVID_WW_TRACE", run with
AVID_HIP_LIB=.../libavid_hip_wwtrace.so): where the waves of wino_wgrad_kernel spend their shader cycles, per phase of the
chunk loop, for the conv2x / conv3x / conv4x spatial layers at batch 64."""
import os, sys, ctypes as C
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "avid-cma_amd"))
import numpy as np, torch
from avid_hip import lib, ops
dev = torch.device("cuda:0")
dll = lib._lib
names = ["k-step 0 second half (+right factor, LDS stores)", "k-step 1 (+loads)", "k-steps 2-3", "barrier", "k-step 0 first half (+left factor)"]
for cin, (T, H, W) in ((64, (8, 28, 28)), (128, (4, 14, 14)), (256, (2, 7, 7))):
    x = torch.randn(64, T, H, W, cin, device=dev).requires_grad_(True)
    w = ops.make_weight(cin, cin, 1, 3, 3).normal_().to(dev).requires_grad_(True)
    y = ops.conv_cl(x, w, (1, 1, 1), (0, 1, 1))
    g = torch.randn_like(y)
    for _ in range(3):
        x.grad = None; w.grad = None
        y.backward(g, retain_graph=True)
    torch.cuda.synchronize()
    lib.timing_enable(True)
    x.grad = None; w.grad = None
    y.backward(g, retain_graph=True)
    torch.cuda.synchronize()
    rep = lib.timing_report(); lib.timing_enable(False)
    us = rep["wino_wgrad_kernel"]["ms"] * 1e3
    buf = np.zeros(1024 * 8 * 8, dtype=np.int64)
    assert dll.avid_debug_ww_trace(buf.ctypes.data_as(C.c_void_p)) == 0
    tr = buf.reshape(1024, 8, 8)
    used = tr[tr.sum(axis=(1, 2)) > 0]
    tot = used[:, :, :5].sum(axis=2)                    # [wg][wave]
    ghz = used[:, :, 6].sum() / (used[:, :, 7].sum() * 10.0)
    print(f"C={cin}: kernel {us:.1f} us, {len(used)} workgroups; cycles per wave in the chunk loop: mean {tot.mean():.0f} (min {tot.min()}, max {tot.max()}); clock in the kernel {ghz:.3f} GHz")
    for role, sl in (("waves 0-3 (V transform)", slice(0, 4)), ("waves 4-7 (dM transform)", slice(4, 8))):
        t = used[:, sl, :5].reshape(-1, 5).mean(axis=0)
        print(f"   {role}: " + ", ".join(f"{n} {v:.0f} ({100 * v / t.sum():.0f} %)" for n, v in zip(names, t)))
