"""Dev tool: the fused cross-modal criterion kernel alone (criterions/avid.py:52-71 + nce.py:38-58) on banks of 240k and
2M rows — time per launch (library HIP events) and the gathered rows' read bandwidth.  AVID_XM_ROWS / AVID_XM_NT select
the variant (csrc/criterion.hip).  usage: python tools/xmodal_bench.py [reps=20]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "avid-cma_amd"))
import torch, criterions
from avid_hip import lib
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda:0")
B, K = 64, 1024
for N in (240000, 2000000):
    crit = criterions.AVID(num_data=N, embedding_dim=128, num_negatives=K, momentum=0.5, device=0)
    ve = torch.randn(B, 128, device=dev, requires_grad=True); ae = torch.randn(B, 128, device=dev, requires_grad=True)
    ids = [torch.randperm(N)[:B].to(dev) for _ in range(reps + 3)]
    if os.environ.get("XB_SEQ") == "1":      # consecutive rows instead of random ones: the kernel's time without the random access
        seq = (torch.arange(B * K, device=dev).view(B, K) * 1 + 7) % (N - 1)
        crit.nce_average.sample_negatives = lambda yy, KK: seq
    for i in range(3):
        loss, _ = crit(ve, ae, ids[i]); loss.backward()
    torch.cuda.synchronize()
    lib.timing_enable(True)
    for i in range(reps):
        loss, _ = crit(ve, ae, ids[3 + i]); loss.backward()
    torch.cuda.synchronize()
    k = lib.timing_report(); lib.timing_enable(False)
    for n in ("xmodal_fused_kernel", "xmodal_finish_kernel", "alias_draw_kernel", "bank_update2_kernel"):
        x = k.get(n)
        if x:
            print(f"N {N:8d} {n:24s} {x['ms']/x['launches']*1e3:7.2f} us/launch" +
                  (f"  {x['bytes']/(x['ms']*1e-3)/1e9:7.0f} GB/s of gathered rows" if x['bytes'] else ""))
    del crit
