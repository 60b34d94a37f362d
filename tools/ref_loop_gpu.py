"""The reference's loop shape (torch DDP on a one-rank group + torch.optim.Adam + loss.item(), main-avid.py:155-180) alone, for
`rocprofv3 --kernel-trace --stats`: which kernels it runs that TrainStep does not (the optimizer's, the reducer's).
  REF_LOOP=0: TrainStep instead (the same number of steps), for the difference of the two kernel tables."""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "avid-cma_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402


def main():
    import models
    import criterions
    import torch.distributed as dist
    from torch.nn.parallel import DistributedDataParallel
    from avid_hip.parallel import TrainStep
    dev = torch.device("cuda:0")
    bs, N, n = 64, 240000, int(os.environ.get("REF_STEPS", "20"))
    torch.manual_seed(0)
    model = models.av_wrapper("R2Plus1D", {"depth": 18}, "Conv2D", {"depth": 10}, proj_dim=[512, 512, 128]).to(dev).train()
    crit = criterions.AVID(num_data=N, embedding_dim=128, num_negatives=1024, momentum=0.5, xModal_coeff=1., wModal_coeff=0., device=0)
    g = torch.Generator().manual_seed(1)
    video = torch.randn(bs, 3, 8, 112, 112, generator=g).to(dev)
    audio = torch.randn(bs, 1, 40, 100, generator=g).to(dev)
    ids = torch.stack([torch.randperm(N, generator=g)[:bs] for _ in range(64)]).to(dev)
    if os.environ.get("REF_LOOP", "1") == "0":
        eng = TrainStep(model, crit)
        one = lambda i: eng.step(video, audio, ids[i % 64])
    else:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29534")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        model.overlap_towers = False
        net = DistributedDataParallel(model, device_ids=[0])
        opt = torch.optim.Adam(model.parameters(), lr=2e-4, betas=(0.9, 0.999), weight_decay=1e-5)

        def one(i):
            v, a = net(video, audio)
            loss, _ = crit(v, a, ids[i % 64])
            loss.item()
            opt.zero_grad()
            loss.backward()
            opt.step()
    for i in range(5):
        one(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        one(5 + i)
    torch.cuda.synchronize()
    print("%.3f ms/step over %d steps" % ((time.perf_counter() - t0) / n * 1e3, n))
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
