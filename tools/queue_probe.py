"""Dev tool: which HIP streams slow each other down?  Streams are created in order (GPU_MAX_HW_QUEUES decides how many
hardware queues they spread over); for every pair (a, b) a chain of tiny kernels runs on a and on b concurrently and the
wall time per kernel is printed relative to a chain running alone."""
import os
import sys
import time
import torch

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
bufs = [torch.zeros(int(os.environ.get('PROBE_ELEMS', 32 << 20)), device=dev) for _ in streams]
N = 120


def chain(idx):
    for k in idx:
        with torch.cuda.stream(streams[k]):
            bufs[k].add_(1.0)


def run(pair):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N):
        chain(pair)
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
