"""Dev tool: what does a short co-runner on another stream cost a persistent convolution kernel?
Each layer's forward is timed alone and with an 85 MB fill (22 us alone) released on a second stream at the same
moment / 30 us earlier / 30-100 us later (both streams wait for one long kernel; the offsets are a fill of known
duration in front of one of them)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "avid-cma_amd"))
import torch
from avid_hip import ops
dev = torch.device("cuda:0")
B = 64
L = [("stem", 3, 64, (3, 7, 7), (1, 2, 2), (1, 3, 3), (8, 112, 112), True),
     ("c2.spt", 64, 64, (1, 3, 3), (1, 1, 1), (0, 1, 1), (8, 28, 28), False),
     ("c2.tmp", 64, 64, (3, 1, 1), (1, 1, 1), (1, 0, 0), (8, 28, 28), False),
     ("c3.spt", 128, 128, (1, 3, 3), (1, 1, 1), (0, 1, 1), (4, 14, 14), False)]
import models
_m = models.av_wrapper("R2Plus1D", {"depth": 18}, "Conv2D", {"depth": 10}, proj_dim=[512, 512, 128]).to(dev).train()
twt = ops.TransposedWeights([p for p in _m.parameters() if p.dim() >= 2])
CO = os.environ.get("CO", "fill")
side = torch.cuda.Stream(dev)
big = torch.zeros(256 << 20, device=dev)                  # the gate: a 1 GB fill (~250 us)
victim = torch.zeros((85 << 20) // 4, device=dev)        # the co-runner's 85 MB
pad = torch.zeros((110 << 20) // 4, device=dev)          # ~30 us of fill: the offset
def timed(fn, mode):
    main = torch.cuda.current_stream()
    ts = []
    for _ in range(6):
        big.zero_()
        gate = torch.cuda.Event(); gate.record()
        if mode is not None:
            with torch.cuda.stream(side):
                side.wait_event(gate)
                for _ in range(max(0, mode)): pad.zero_()          # co-runner later by mode x 30 us
                if CO == "fill": victim.zero_()
                else: twt.refresh()
        for _ in range(max(0, -(mode or 0))): pad.zero_()          # co-runner earlier
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        main.wait_stream(side)
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return sorted(ts)[len(ts) // 2]
print(f"{'layer':8s} {'alone':>8s} {'same t':>8s} {'-30us':>8s} {'+30us':>8s} {'+90us':>8s}   (us; co-runner: 85 MB fill)")
for name, cin, cout, k, st, pd, (T, H, W), cf in L:
    x = torch.randn((B, cin, T, H, W) if cf else (B, T, H, W, cin), device=dev)
    w = ops.make_weight(cout, cin, *k).normal_().to(dev)
    fn = lambda: ops.conv_cl(x, w, st, pd, channel_first=cf)
    fn(); torch.cuda.synchronize()
    r = [timed(fn, m) for m in (None, 0, -1, 1, 3)]
    print(f"{name:8s} " + " ".join(f"{v:8.1f}" for v in r))
