"""Dev tool: cost of a kernel boundary in one HIP stream (dependent launches), eager and graph-replayed."""
import torch, time
dev = torch.device("cuda:0")
x = torch.zeros(64, device=dev)
big = torch.zeros(64 << 20, device=dev)   # 256 MB: ~100 us kernels
def run(n, f):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e6
tiny = lambda: x.add_(1.0)
print(f"eager tiny kernel chain: {run(2000, tiny):.2f} us/launch")
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(1000): x.add_(1.0)
print(f"graph tiny kernel chain: {run(5, g.replay)/1000:.2f} us/kernel")
# big kernel alone vs big + tiny alternating (graph): the extra per pair is boundary + tiny kernel
g1 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g1):
    for _ in range(100): big.add_(1.0)
g2 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g2):
    for _ in range(100): big.add_(1.0); x.add_(1.0)
a = run(5, g1.replay) / 100; b = run(5, g2.replay) / 100
print(f"big alone {a:.2f} us; big+tiny {b:.2f} us -> extra {b-a:.2f} us per inserted tiny kernel")
