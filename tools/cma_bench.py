"""Dev tool: time the CMA correspondence search (criterions/avid_cma.py:42-123) on one GPU — a slice of the
query rows of an N x 128 bank pair, extrapolated to the whole bank (SURVEY §8d cfg 4: 29.5 TFLOP at N = 240k)."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "avid-cma_amd"))
import torch
from avid_hip import topk, lib
dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 240000
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
g = torch.Generator().manual_seed(0)
V = torch.nn.functional.normalize(torch.randn(N, 128, generator=g), dim=1).to(dev)
A = torch.nn.functional.normalize(torch.randn(N, 128, generator=g), dim=1).to(dev)
for batch in (256, 1024):
    topk.cma_topk(V, A, 0, 2 * batch, 32, 0, batch); torch.cuda.synchronize()
    lib.timing_enable(True)
    t0 = time.perf_counter()
    out = topk.cma_topk(V, A, 0, nq, 32, 0, batch)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    rep = lib.timing_report(); lib.timing_enable(False)
    fl = 4.0 * N * nq * 128
    print(f"N={N} queries={nq} batch={batch}: {dt*1e3:8.1f} ms  {fl/dt/1e12:6.1f} TFLOP/s  -> whole bank {dt*N/nq:6.2f} s on one GPU")
    for k, v in sorted(rep.items(), key=lambda kv: -kv[1]["ms"])[:4]:
        print(f"      {k:34s} {v['ms']:8.1f} ms")
