"""Where the reference's loop with the drop-in objects (avid_hip.parallel.DistributedDataParallel + Adam) loses against TrainStep.step:
the same loop with / without loss.item(), with / without a one-rank RCCL group (AVID_FORCE_DIST=1), ms per step each."""
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
    from avid_hip import parallel
    dev = torch.device("cuda:0")
    bs, N, n = 64, 240000, int(os.environ.get("GAP_STEPS", "60"))
    g = torch.Generator().manual_seed(1)
    video = torch.randn(bs, 3, 8, 112, 112, generator=g).to(dev)
    audio = torch.randn(bs, 1, 40, 100, generator=g).to(dev)
    ids = torch.stack([torch.randperm(N, generator=g)[:bs] for _ in range(64)]).to(dev)

    def fresh():
        torch.manual_seed(0)
        m = models.av_wrapper("R2Plus1D", {"depth": 18}, "Conv2D", {"depth": 10}, proj_dim=[512, 512, 128]).to(dev).train()
        c = criterions.AVID(num_data=N, embedding_dim=128, num_negatives=1024, momentum=0.5, xModal_coeff=1., wModal_coeff=0., device=0)
        return m, c

    def timed(one, label):
        for i in range(8):
            one(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            one(8 + i)
        torch.cuda.synchronize()
        print("%-64s %.3f ms/step" % (label, (time.perf_counter() - t0) / n * 1e3), flush=True)

    def loop(item, phases=None):
        m, c = fresh()
        net = parallel.DistributedDataParallel(m, device_ids=[0])
        opt = parallel.Adam(net.parameters(), lr=2e-4, betas=(0.9, 0.999), weight_decay=1e-5)

        def one(i):
            t0 = time.perf_counter()
            v, a = net(video, audio)
            loss, _ = c(v, a, ids[i % 64])
            t1 = time.perf_counter()
            if item:
                loss.item()
            t2 = time.perf_counter()
            opt.zero_grad()
            t3 = time.perf_counter()
            loss.backward()
            t4 = time.perf_counter()
            opt.step()
            t5 = time.perf_counter()
            if phases is not None and i >= 8:
                for k, d in zip(("fwd+crit", "item", "zero", "bwd", "opt"), (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)):
                    phases[k] = phases.get(k, 0.0) + d
        return one

    for rep in range(2):
        m, c = fresh()
        eng = parallel.TrainStep(m, c)
        timed(lambda i: eng.step(video, audio, ids[i % 64]), "TrainStep.step")
        timed(lambda i: (eng.step(video, audio, ids[i % 64]).item()), "TrainStep.step + loss.item() behind the step")
        del eng
        ph = {}
        timed(loop(True, ph), "drop-in loop")
        print("    host ms: " + "  ".join(f"{k} {v / n * 1e3:.3f}" for k, v in ph.items()))
        timed(loop(False), "drop-in loop without loss.item()")
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", "29541"
    os.environ["AVID_FORCE_DIST"] = "1"
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    timed(loop(True), "drop-in loop, one-rank RCCL group")
    timed(loop(False), "drop-in loop, one-rank RCCL group, without loss.item()")
    m, c = fresh()
    eng = parallel.TrainStep(m, c)
    timed(lambda i: eng.step(video, audio, ids[i % 64]), "TrainStep.step, one-rank RCCL group")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
