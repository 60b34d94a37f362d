"""Dev tool: the other BASELINE.json configs on ONE GPU (the driver's bench.py run is config 2):
  config 4: AVID+CMA InstX-N1024-PosW-N64-Top32 step (criterions.AVID_CMA; its constructor runs the top-K search)
  config 5: AVID step with an Audioset-scale 2M x 128 bank
usage: python tools/config_bench.py [cma|bank2m|both] [steps=20]"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "avid-cma_amd"))
import torch, models, criterions
from avid_hip.parallel import TrainStep
which = sys.argv[1] if len(sys.argv) > 1 else "both"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda:0")
B = 64
v = torch.randn(B, 3, 8, 112, 112, device=dev); a = torch.randn(B, 1, 40, 100, device=dev)

def run(name, crit, N):
    m = models.av_wrapper("R2Plus1D", {"depth": 18}, "Conv2D", {"depth": 10}, proj_dim=[512, 512, 128]).to(dev).train()
    e = TrainStep(m, crit)
    ids = [torch.randperm(N)[:B].to(dev) for _ in range(steps + 5)]
    for i in range(5): e.step(v, a, ids[i])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(steps): loss = e.step(v, a, ids[5 + i])
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
    print(f"{name}: {dt*1e3:.2f} ms/step  {B/dt:.0f} clips/s  loss {float(loss):.4f}  "
          f"(HBM in use {torch.cuda.memory_allocated()/2**30:.1f} GiB)")
    from avid_hip import lib
    m.overlap_towers = False
    lib.timing_enable(True)
    for i in range(3): e.step(v, a, ids[i])
    torch.cuda.synchronize()
    k = lib.timing_report(); lib.timing_enable(False)
    for n, x in k.items():
        if n.startswith(("bank_", "nce_", "alias", "cma_neg")):
            print(f"    {n:28s} {x['launches']/3:4.1f}/step {x['ms']/x['launches']*1e3:7.1f} us/launch  "
                  f"{x['bytes']/(x['ms']*1e-3)/1e9:7.1f} GB/s (algorithmic bytes)")

if which in ("cma", "both"):
    N = 240000
    torch.cuda.synchronize(); t0 = time.perf_counter()
    c = criterions.AVID_CMA(num_data=N, embedding_dim=128, num_negatives=1024, num_negatives_within=64, momentum=0.5,
                            xModalInstCoeff=1., wModalInstCoeff=0., xModalPosCoeff=0., wModalPosCoeff=1.,
                            sampling_args={"type": "consensus", "pos_k": 32}, resample_freq=-1, device=0)
    torch.cuda.synchronize()
    print(f"AVID_CMA constructor incl. top-32 consensus search over {N} rows: {time.perf_counter()-t0:.2f} s")
    run("config 4 (AVID+CMA InstX-N1024-PosW-N64-Top32, 240k bank)", c, N)
    del c
if which in ("bank2m", "both"):
    N = 2000000
    c = criterions.AVID(num_data=N, embedding_dim=128, num_negatives=1024, momentum=0.5, device=0)
    run("config 5 (AVID Cross-N1024, 2M x 128 bank)", c, N)
