"""Dev tool: how far ahead of the GPU is the host at the phase boundaries of an eager step?  Host timestamps and CUDA
events at: step start, model forward issued, criterion issued, backward issued, optimizer issued.  lead = (time the GPU
reaches the point) - (time the host enqueued it): near zero means the GPU is waiting for the host there."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "avid-cma_amd"))
import torch, models, criterions
from avid_hip.parallel import TrainStep
dev = torch.device("cuda:0")
m = models.av_wrapper("R2Plus1D", {"depth": 18}, "Conv2D", {"depth": 10}, proj_dim=[512, 512, 128]).to(dev).train()
c = criterions.AVID(num_data=240000, embedding_dim=128, num_negatives=1024, momentum=0.5, device=0)
e = TrainStep(m, c)
if os.environ.get('DUMMY_MB'): dummy = torch.empty(int(float(os.environ['DUMMY_MB']) * (1 << 20)), dtype=torch.uint8, device=dev)
v = torch.randn(64, 3, 8, 112, 112, device=dev); a = torch.randn(64, 1, 40, 100, device=dev)
y = torch.randperm(240000)[:64].to(dev)
marks = []
def mark(name):
    ev = torch.cuda.Event(enable_timing=True); ev.record(); marks.append((name, time.perf_counter(), ev))
mf, cf, fb, os_ = m.forward, c.forward, e.forward_backward, e.optimizer_step
def mfw(*a_, **k): r = mf(*a_, **k); mark("forward"); return r
def cfw(*a_, **k): r = cf(*a_, **k); mark("criterion"); return r
def fbw(*a_, **k): r = fb(*a_, **k); mark("backward"); return r
def osw(*a_, **k): r = os_(*a_, **k); mark("optimizer"); return r
m.forward, c.forward, e.forward_backward, e.optimizer_step = mfw, cfw, fbw, osw
for _ in range(4): e.step(v, a, y)
torch.cuda.synchronize(); marks.clear()
N = 6
mark("start")
for _ in range(N): e.step(v, a, y)
torch.cuda.synchronize()
h0, e0 = marks[0][1], marks[0][2]
print(f"{'point':12s} {'host ms':>9s} {'gpu ms':>9s} {'lead ms':>8s} {'host d':>7s} {'gpu d':>7s}")
ph = pg = 0.0
for name, h, ev in marks[1:]:
    hm, gm = (h - h0) * 1e3, e0.elapsed_time(ev)
    print(f"{name:12s} {hm:9.3f} {gm:9.3f} {gm-hm:8.3f} {hm-ph:7.3f} {gm-pg:7.3f}")
    ph, pg = hm, gm
