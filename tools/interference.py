"""Dev tool: how much does a long-lived foreign kernel (a stand-in for RCCL's all-reduce workgroups) cost the
training step's persistent kernels?  Launches `interfere_kernel` (tools/interfere.hip) on its own stream at the start
of every backward pass and compares clips/s with and without it.
usage: python tools/interference.py [G=32] [threads=512] [lds_kb=16] [us=1500]"""
import ctypes as C, os, subprocess, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "avid-cma_amd"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch, models, criterions
from avid_hip.parallel import TrainStep
so = os.path.join(REPO, "tools", "bin", "libinterfere.so")
if not os.path.exists(so):
    os.makedirs(os.path.dirname(so), exist_ok=True)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-o", so,
                           os.path.join(REPO, "tools", "interfere.hip")])
lib = C.CDLL(so)
lib.interfere_launch.argtypes = [C.c_void_p, C.c_longlong, C.c_int, C.c_int, C.c_int, C.c_double, C.c_void_p]
G, T, lds_kb, us = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((1, 32), (2, 512), (3, 16), (4, 1500)))
dev = torch.device("cuda:0")
m = models.av_wrapper("R2Plus1D", {"depth": 18}, "Conv2D", {"depth": 10}, proj_dim=[512, 512, 128]).to(dev).train()
c = criterions.AVID(num_data=240000, embedding_dim=128, num_negatives=1024, momentum=0.5, device=0)
e = TrainStep(m, c)
v = torch.randn(64, 3, 8, 112, 112, device=dev); a = torch.randn(64, 1, 40, 100, device=dev)
y = torch.randperm(240000)[:64].to(dev)
buf = torch.zeros(64 << 20, device=dev)
side = torch.cuda.Stream(dev)
orig_backward = torch.Tensor.backward

def run(interfere, steps=30):
    def hooked(self, *a_, **k_):
        if interfere:
            side.wait_stream(torch.cuda.current_stream())
            lib.interfere_launch(buf.data_ptr(), buf.numel(), G, T, lds_kb << 10, float(us), C.c_void_p(side.cuda_stream))
        return orig_backward(self, *a_, **k_)
    torch.Tensor.backward = hooked
    try:
        for _ in range(5): e.step(v, a, y)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(steps): e.step(v, a, y)
        torch.cuda.synchronize()
        return 64 * steps / (time.perf_counter() - t0)
    finally:
        torch.Tensor.backward = orig_backward

base = run(False)
with_k = run(True)
print(f"G={G} x {T} threads, {lds_kb} KB LDS, {us} us per backward: {base:.0f} -> {with_k:.0f} clips/s ({100*(with_k/base-1):+.1f} %)")
