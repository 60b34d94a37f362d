"""Helper of tests/test_gpu_switches.py: two engine steps on a fixed small batch in a fresh process (environment switches
are read once at import / first use), one JSON line with the losses and a digest of the last gradient buffer."""
import hashlib
import json
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "avid-cma_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
from oracle import detgen  # noqa: E402


def main():
    import models
    import criterions
    from avid_hip.parallel import TrainStep
    dev = torch.device("cuda:0")
    if os.environ.get("AVID_FORCE_DIST") == "1":
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29561")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    m = models.av_wrapper("R2Plus1D", {"depth": 18}, "Conv2D", {"depth": 10}, proj_dim=[512, 512, 128])
    sd = m.state_dict()
    m.load_state_dict({k: torch.from_numpy(detgen.det_param(f"w:{k}", tuple(v.shape)).copy()).to(v.dtype) for k, v in sd.items()})
    m = m.to(dev).train()
    N, K, bs = 2000, 128, int(os.environ.get("PROBE_BS", "4"))
    crit = criterions.AVID(num_data=N, embedding_dim=128, num_negatives=K, momentum=0.5, device=0)
    gg = torch.Generator().manual_seed(3)
    crit.nce_average.view1_mem.copy_(F.normalize(torch.randn(N, 128, generator=gg), dim=1))
    crit.nce_average.view2_mem.copy_(F.normalize(torch.randn(N, 128, generator=gg), dim=1))
    crit.nce_average.multinomial.reseed(11, 0)
    eng = TrainStep(m, crit)
    g = torch.Generator().manual_seed(5)
    video = torch.randn(bs, 3, 8, 112, 112, generator=g).to(dev)
    audio = torch.randn(bs, 1, 40, 100, generator=g).to(dev)
    ids = torch.stack([torch.randperm(N, generator=g)[:bs] for _ in range(3)]).to(dev)
    losses = []
    for i in range(3):
        loss = eng.forward_backward(video, audio, ids[i])
        torch.cuda.synchronize()
        if i == 0:
            grad = eng.flat.grad.clone()          # the FIRST step's gradients: same weights in every setting
        eng.optimizer_step()
        losses.append(float(loss.detach()))
    gn = grad.double()
    out = {"losses": losses, "grad_norm": float(gn.norm()), "grad_sha": hashlib.sha256(grad.cpu().numpy().tobytes()).hexdigest(),
           "grad_probe": [float(v) for v in grad[:: max(1, grad.numel() // 64)][:64].cpu()]}
    print("PROBE " + json.dumps(out), flush=True)
    if os.environ.get("AVID_FORCE_DIST") == "1":
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
