"""Dev tool: the bs = 64 full step on the device vs the oracle, every parameter gradient's error listed.
    python tools/bs64_debug.py oracle   # device run (current env) + oracle with its masks -> /tmp/bs64_oracle.pt, table
    python tools/bs64_debug.py dev      # device run under the current env vs the saved oracle gradients
"""
import os, sys
import numpy as np
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "avid-cma_amd"))
from oracle import avid_oracle as O, detgen
from oracle.hooks import capture_relu_masks, capture_pool_argmax, pool_pick_report
import models, criterions

T = lambda a: torch.from_numpy(np.ascontiguousarray(a))
dev = torch.device("cuda:0")
bs = int(os.environ.get("DBG_BS", "64"))
N, K = 240_000, 1024


def inputs():
    g = torch.Generator().manual_seed(20260928)
    video = torch.randn(bs, 3, 8, 112, 112, generator=g)
    audio = torch.randn(bs, 1, 40, 100, generator=g)
    y = torch.randperm(N, generator=g)[:bs]
    idx = torch.randint(0, N - 1, (bs, K), generator=g)
    idx = idx + (idx >= y[:, None]).long()
    v1 = torch.nn.functional.normalize(torch.randn(N, 128, generator=g), dim=1)
    v2 = torch.nn.functional.normalize(torch.randn(N, 128, generator=g), dim=1)
    return video, audio, y, idx, v1, v2


def device_run(video, audio, y, idx, v1, v2, want_masks):
    P = O.det_state(O.av_wrapper_spec(18), "w")
    m = models.av_wrapper("R2Plus1D", {"depth": 18}, "Conv2D", {"depth": 10}, proj_dim=[512, 512, 128])
    m.load_state_dict({k: v.clone() for k, v in P.items()})
    m = m.to(dev).train()
    crit = criterions.AVID(num_data=N, embedding_dim=128, num_negatives=K, momentum=0.5, device=0)
    crit.nce_average.view1_mem.copy_(v1)
    crit.nce_average.view2_mem.copy_(v2)
    idx_d = idx.to(dev)
    crit.nce_average.sample_negatives = lambda yy, KK: idx_d
    masks, remove = capture_relu_masks(m) if want_masks else ({}, lambda: None)
    picks, remove2 = capture_pool_argmax(m)
    e1, e2 = m(video.to(dev), audio.to(dev))
    remove(); remove2()
    masks["__picks__"] = picks
    loss, _ = crit(e1, e2, y.to(dev))
    loss.backward()
    torch.cuda.synchronize()
    return float(loss), {n: p.grad.contiguous().cpu() for n, p in m.named_parameters()}, masks


def table(grads, ref, thr=1e-4):
    rows = []
    for n in grads:
        a, r = grads[n].double(), ref[n].double()
        rows.append((float((a - r).abs().max() / (r.abs().max() + 1e-30)), n, int(((a - r).abs() > thr * r.abs().max()).sum()), a.numel()))
    rows.sort(reverse=True)
    for e, n, bad, tot in rows[:25]:
        print(f"{e:10.3e}  {n:50s} entries off by > {thr:g} of scale: {bad} / {tot}")
    print("worst", rows[0][:2], " count > 5e-4:", sum(1 for r in rows if r[0] > 5e-4))


if __name__ == "__main__":
    mode = sys.argv[1]
    inp = inputs()
    if mode == "oracle":
        loss, grads, masks = device_run(*inp, True)
        video, audio, y, idx, v1, v2 = inp
        P = O.det_state(O.av_wrapper_spec(18), "w")
        for n in P:
            if not ("running" in n or "num_batches" in n):
                P[n].requires_grad_(True)
        picks = masks.pop("__picks__")
        O.POOL_INPUT = {}
        with torch.no_grad():
            O.av_forward(video, audio, {k: v.detach().clone() for k, v in P.items()}, 18, True)
        print("pool picks vs free oracle (disagree, total, worst gap / rms):", pool_pick_report(picks, O.POOL_INPUT))
        O.POOL_INPUT = None
        O.RELU_MASKS, O.POOL_ARGMAX = masks, picks
        ve, ae = O.av_forward(video, audio, P, 18, True)
        ref_loss, _, _ = O.avid_forward(ve, ae, y, idx, v1.clone(), v2.clone(), None, 0.5)
        ref_loss.backward()
        O.RELU_MASKS = O.POOL_ARGMAX = None
        ref = {n: P[n].grad.clone() for n in grads}
        torch.save({"ref": ref, "loss": float(ref_loss)}, "/tmp/bs64_oracle.pt")
        print("loss", loss, float(ref_loss))
        table(grads, ref)
    else:
        sv = torch.load("/tmp/bs64_oracle.pt")
        loss, grads, _ = device_run(*inp, False)
        print("env", {k: v for k, v in os.environ.items() if k.startswith("AVID_")}, "loss", loss, sv["loss"])
        table(grads, sv["ref"])
