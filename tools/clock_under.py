"""Dev tool: the shader clock while one kernel runs back to back (avid_clock_probe on a second stream)."""
import os, sys, ctypes as C
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/avid-cma_amd")
import torch
from avid_hip import ops, lib, streams
dev = torch.device("cuda:0")
x = torch.randn(64, 3, 8, 112, 112, device=dev)
w = ops.make_weight(64, 3, 3, 7, 7).to(dev)
w.copy_(torch.randn(64, 3, 3, 7, 7, device=dev) * 0.05)
ss = streams.place(dev)
out = torch.zeros(2, dtype=torch.int64, device=dev)
def run(n):
    for _ in range(n):
        ops.conv_cl(x, w, (1, 2, 2), (1, 3, 3), channel_first=True, bn_stats=True)
run(5); torch.cuda.synchronize()
lib.call("avid_clock_probe", 20000, C.c_void_p(out.data_ptr()), C.c_void_p(ss.comm.cuda_stream))
run(40)
torch.cuda.synchronize()
o = out.tolist()
print("clock under the stem forward (split=%s): %.3f GHz" % (os.environ.get("AVID_STEM_BF16X3", "1"), o[0] / o[1] / 10))
torch.cuda.synchronize()
lib.call("avid_clock_probe", 20000, C.c_void_p(out.data_ptr()), C.c_void_p(ss.comm.cuda_stream))
torch.cuda.synchronize()
o = out.tolist()
print("clock idle: %.3f GHz" % (o[0] / o[1] / 10))
