"""Dev tool: the step engine on a ONE-rank RCCL group (the only multi-GPU fact measurable on a one-GPU box): ms per step
for a distributed arrangement.  usage: [AVID_PLAN=0/1] [GPU_MAX_HW_QUEUES=n] python tools/dist1.py <dist|none> [steps]"""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "avid-cma_amd")):
    sys.path.insert(0, p)
mode = sys.argv[1] if len(sys.argv) > 1 else "early"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
if mode != "none":
    os.environ["AVID_FORCE_DIST"] = "1"
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    import models
    import criterions
    from avid_hip.parallel import TrainStep
    dev = torch.device("cuda:0")
    if mode != "none":
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    bs, N = 64, 240000
    torch.manual_seed(0)
    model = models.av_wrapper("R2Plus1D", {"depth": 18}, "Conv2D", {"depth": 10}, proj_dim=[512, 512, 128]).to(dev).train()
    crit = criterions.AVID(num_data=N, embedding_dim=128, num_negatives=1024, momentum=0.5, xModal_coeff=1., wModal_coeff=0., device=0)
    X = os.environ.get("AVID_X", "")
    eng = TrainStep(model, crit, broadcast_buffers="off" if "nobcast" in X else "step")
    if "noseg" in X:
        eng.buckets._capturing = lambda: True          # one piece, every bucket from finish()... (capture path: one collective)
    if "noar" in X:
        import torch.distributed as d2
        class W:
            def wait(self): pass
        d2.all_reduce = lambda *a, **k: W()
    g = torch.Generator().manual_seed(1)
    video = torch.randn(bs, 3, 8, 112, 112, generator=g).to(dev)
    audio = torch.randn(bs, 1, 40, 100, generator=g).to(dev)
    ids = torch.stack([torch.randperm(N, generator=g)[:bs] for _ in range(8)]).to(dev)
    nd = int(os.environ.get("AVID_DUMMY", "0"))
    dummies = [torch.cuda.Stream(dev) for _ in range(nd)]
    dz = torch.zeros(16, device=dev)

    def poke(all_=False):
        for st in dummies:
            with torch.cuda.stream(st):
                dz.add_(1)
    poke()
    for i in range(8):
        eng.step(video, audio, ids[i % 8])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        eng.step(video, audio, ids[i % 8])
        if os.environ.get("AVID_DUMMY_EVERY"):
            poke()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    print(f"mode={mode} plan={os.environ.get('AVID_PLAN', '1')} hwq={os.environ.get('GPU_MAX_HW_QUEUES', 'default(8 via lib)')} "
          f"extra={os.environ.get('AVID_X', '')}: {ms:.3f} ms/step")
    if "timing" in X:
        from avid_hip import lib
        lib.timing_enable(True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(3):
            eng.step(video, audio, ids[i % 8])
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / 3 * 1e3
        rep = lib.timing_report()
        lib.timing_enable(False)
        tot = sum(v["ms"] for v in rep.values()) / 3
        print(f"   timed pass: wall {wall:.3f} ms/step, kernels {tot:.3f} ms/step, launches {sum(v['launches'] for v in rep.values()) / 3:.0f}")
        for k, v in sorted(rep.items(), key=lambda kv: -kv[1]["ms"])[:8]:
            print(f"      {k:40s} {v['launches'] / 3:6.1f} {v['ms'] / 3:8.3f}")
    if mode != "none":
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
