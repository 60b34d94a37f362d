"""Host time of a step, phase by phase (wall clock of the issuing thread, GPU left to run behind it):
model forward program, criterion, backward (criterion backward + model backward program), Adam — for the step engine,
and for the reference's loop shape (DDP + torch.optim.Adam + loss.item())."""
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
    from avid_hip import plan, lib
    from avid_hip.parallel import TrainStep
    dev = torch.device("cuda:0")
    bs, N = 64, 240000
    torch.manual_seed(0)
    model = models.av_wrapper("R2Plus1D", {"depth": 18}, "Conv2D", {"depth": 10}, proj_dim=[512, 512, 128]).to(dev).train()
    crit = criterions.AVID(num_data=N, embedding_dim=128, num_negatives=1024, momentum=0.5, xModal_coeff=1., wModal_coeff=0., device=0)
    eng = TrainStep(model, crit)
    g = torch.Generator().manual_seed(1)
    video = torch.randn(bs, 3, 8, 112, 112, generator=g).to(dev)
    audio = torch.randn(bs, 1, 40, 100, generator=g).to(dev)
    ids = torch.stack([torch.randperm(N, generator=g)[:bs] for _ in range(64)]).to(dev)
    for i in range(8):
        eng.step(video, audio, ids[i])
    torch.cuda.synchronize()
    T = {"fwd": 0.0, "crit": 0.0, "bwd": 0.0, "finish": 0.0, "adam": 0.0, "poll": 0.0}
    n = 20
    t_all = time.perf_counter()
    for i in range(n):
        t0 = time.perf_counter()
        with plan.engine(eng):
            out = plan.run(model, video, audio)
            t1 = time.perf_counter()
            loss, _ = crit(out[0], out[1], ids[8 + i])
            t2 = time.perf_counter()
            loss.backward()
        t3 = time.perf_counter()
        eng.buckets.finish()
        t4 = time.perf_counter()
        eng.optimizer_step()
        t5 = time.perf_counter()
        eng._poll_errors()
        t6 = time.perf_counter()
        for k, v in zip(T, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5)):
            T[k] += v
    host = time.perf_counter() - t_all
    torch.cuda.synchronize()
    wall = time.perf_counter() - t_all
    print("engine: host %.3f ms/step issue, %.3f ms/step wall" % (host / n * 1e3, wall / n * 1e3))
    print("   " + "  ".join(f"{k} {v / n * 1e3:.3f}" for k, v in T.items()))
    pl = [p for p in model._avid_plans.values() if p][0]
    print("   records: fwd %d bwd %d" % (pl.n_fwd, pl.n_bwd))
    # raw executor cost: the forward program issued 20 times back to back
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(10):
        with plan.engine(eng):
            out = plan.run(model, video, audio)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print("   forward program alone: %.3f ms host per call" % ((t1 - t0) / 10 * 1e3))
    # ---- the reference loop
    from torch.nn.parallel import DistributedDataParallel
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    for use_ddp in (False, True):
        if use_ddp:
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        net = DistributedDataParallel(model, device_ids=[0]) if use_ddp else model
        opt = torch.optim.Adam(model.parameters(), lr=2e-4, betas=(0.9, 0.999), weight_decay=1e-5)
        R = {"fwd": 0.0, "crit": 0.0, "item": 0.0, "zero": 0.0, "bwd": 0.0, "opt": 0.0}

        def one(i, rec):
            t0 = time.perf_counter()
            v, a = net(video, audio)
            t1 = time.perf_counter()
            loss, _ = crit(v, a, ids[i % 64])
            t2 = time.perf_counter()
            loss.item()
            t3 = time.perf_counter()
            opt.zero_grad()
            t4 = time.perf_counter()
            loss.backward()
            t5 = time.perf_counter()
            opt.step()
            t6 = time.perf_counter()
            if rec:
                for k, v_ in zip(R, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5)):
                    R[k] += v_
        for i in range(5):
            one(i, False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            one(i, True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("reference loop (%s): %.3f ms/step" % ("DDP" if use_ddp else "no DDP", dt / n * 1e3))
        print("   " + "  ".join(f"{k} {v / n * 1e3:.3f}" for k, v in R.items()))
        del net, opt
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
