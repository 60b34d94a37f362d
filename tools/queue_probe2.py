"""Dev tool: pair matrix of concurrently spinning single-wave kernels over freshly created streams (see queue_probe.py)."""
import os
import sys
import time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "avid-cma_amd")):
    sys.path.insert(0, p)
import torch
from avid_hip import lib
import ctypes as C

n_streams = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda:0")
if os.environ.get("PROBE_PG"):
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29547")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    t = torch.zeros(1024, device=dev)
    dist.all_reduce(t)
    torch.cuda.synchronize()
streams = [torch.cuda.current_stream(dev)] + [torch.cuda.Stream(dev) for _ in range(n_streams)]
h = [C.c_void_p(s.cuda_stream) for s in streams]
N, US = 30, 30


def run(pair):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N):
        for k in pair:
            lib.call("avid_probe_spin", US, h[k])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / N * 1e6


for k in range(len(streams)):
    run((k,))
alone = [run((k,)) for k in range(len(streams))]
print("alone us/kernel:", " ".join(f"{v:5.1f}" for v in alone))
print("pair matrix (us per round of two kernels):")
for a in range(len(streams)):
    row = []
    for b in range(len(streams)):
        row.append("   . " if b <= a else f"{run((a, b)):5.1f}")
    print(f"  s{a:2d}: " + " ".join(row))
