"""Dev tool: on which stream does torch's NCCL process group run a stream-synchronous collective?  One-rank group,
out-of-place all_gather (a device copy) issued under a side stream, synchronous and asynchronous; run under
`rocprofv3 --kernel-trace` and compare the Stream_Id / Queue_Id of the copy kernels with the marker kernels."""
import os
import sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "avid-cma_amd")):
    sys.path.insert(0, p)
import ctypes as C
import torch
import torch.distributed as dist
from avid_hip import lib

dev = torch.device("cuda:0")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29549")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
s = torch.cuda.Stream(dev)
inp = torch.ones(1 << 20, device=dev)
out = torch.zeros(1 << 20, device=dev)
torch.cuda.synchronize()
lib.call("avid_probe_spin", 7, C.c_void_p(torch.cuda.current_stream().cuda_stream))     # marker: main, 7 us
with torch.cuda.stream(s):
    lib.call("avid_probe_spin", 13, C.c_void_p(s.cuda_stream))                          # marker: side, 13 us
    for _ in range(3):
        dist.all_gather_into_tensor(out, inp)                                           # synchronous op under s
torch.cuda.synchronize()
lib.call("avid_probe_spin", 7, C.c_void_p(torch.cuda.current_stream().cuda_stream))
with torch.cuda.stream(s):
    ws = [dist.all_gather_into_tensor(out, inp, async_op=True) for _ in range(3)]
    for w in ws:
        w.wait()
torch.cuda.synchronize()
dist.destroy_process_group()
