#!/usr/bin/env python
"""Generate tests/golden/*.npz by IMPORTING the reference (runs only in the build container).

    python tools/make_golden.py [--ref /root/reference] [--out tests/golden]

The reference hard-codes ``.cuda()`` in its criterions (criterions/avid.py:93,96,179;
criterions/avid_cma.py:224,294); in THIS process only we neutralise ``Tensor.cuda`` /
``Module.cuda`` before importing it.  No reference file is edited or copied; only inputs
(regenerable from oracle/detgen.py) and the reference's numeric outputs are stored.
"""
import argparse
import os
import queue
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle import detgen  # noqa: E402
from oracle import avid_oracle as O  # noqa: E402 (only for *_spec name lists and det_state)


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def load_det_weights(module, tag):
    sd = module.state_dict()
    new = {k: T(detgen.det_param(f"{tag}:{k}", tuple(v.shape))).to(v.dtype) for k, v in sd.items()}
    module.load_state_dict(new)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(REPO, "tests", "golden"))
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)

    torch.Tensor.cuda = lambda s, *a, **k: s
    torch.nn.Module.cuda = lambda s, *a, **k: s
    sys.path.insert(0, args.ref)
    import models  # reference
    import criterions  # reference
    from criterions.nce import NCECriterion
    from criterions.avid_cma import CMASampler
    from models.network_blocks import BasicR2P1DBlock, Basic2DBlock
    from utils.alias_method import AliasMethod

    torch.manual_seed(0)
    torch.set_num_threads(8)

    # ---------------------------------------------------------------- 1. alias tables
    out = {}
    for name, probs in [("ones999", np.ones(999, np.float32)),
                        ("p4", np.array([.5, .3, .1, .1], np.float32)),
                        ("det50", np.abs(detgen.det_uniform("alias:det50", (50,))) + 0.01)]:
        am = AliasMethod(T(probs.copy()))
        out[f"{name}_probs"] = probs
        out[f"{name}_prob"] = am.prob.numpy()
        out[f"{name}_alias"] = am.alias.numpy()
    np.savez_compressed(os.path.join(args.out, "alias.npz"), **out)

    # ---------------------------------------------------------------- 2. NCECriterion
    out = {}
    for tag, (bs, Pn, K) in {"p1k64": (4, 1, 64), "p32k64": (3, 32, 64)}.items():
        sp = T(detgen.det_uniform(f"nce:{tag}:pos", (bs, Pn)) * 8.0).requires_grad_(True)
        sn = T(detgen.det_uniform(f"nce:{tag}:neg", (bs, K)) * 8.0).requires_grad_(True)
        crit = NCECriterion(1000)
        loss = crit(sp, sn)
        loss.backward()
        out[f"{tag}_loss1"], out[f"{tag}_Z"] = loss.item(), float(crit.avg_exp_score)
        out[f"{tag}_gpos1"], out[f"{tag}_gneg1"] = sp.grad.numpy().copy(), sn.grad.numpy().copy()
        sp.grad = None; sn.grad = None
        sp2 = (sp.detach() * 0.5).requires_grad_(True)
        loss2 = crit(sp2, sn)          # stored Z reused
        loss2.backward()
        out[f"{tag}_loss2"] = loss2.item()
        out[f"{tag}_gpos2"], out[f"{tag}_gneg2"] = sp2.grad.numpy().copy(), sn.grad.numpy().copy()
    np.savez_compressed(os.path.join(args.out, "nce.npz"), **out)

    # ---------------------------------------------------------------- 3. AVID (injected idx)
    def det_bank(tag, N, D=128):
        b = T(detgen.det_normalish(f"bank:{tag}", (N, D)))
        return torch.nn.functional.normalize(b, p=2, dim=1)

    out = {}
    for tag, (N, bs, K, xc, wc) in {"cross": (1000, 4, 64, 1.0, 0.0), "joint": (300, 3, 16, 1.0, 1.0)}.items():
        crit = criterions.AVID(num_data=N, embedding_dim=128, num_negatives=K, momentum=0.5,
                               xModal_coeff=xc, wModal_coeff=wc)
        crit.nce_average.view1_mem.copy_(det_bank(f"{tag}:v1", N))
        crit.nce_average.view2_mem.copy_(det_bank(f"{tag}:v2", N))
        for step in range(2):
            v = T(detgen.det_normalish(f"avid:{tag}:v{step}", (bs, 128))).requires_grad_(True)
            a = T(detgen.det_normalish(f"avid:{tag}:a{step}", (bs, 128))).requires_grad_(True)
            y = T(detgen.det_indices(f"avid:{tag}:y{step}", bs, N))
            if step == 1:
                y[0] = out[f"{tag}_y0"][1]      # revisit a row updated in step 0
            draw = T(detgen.det_indices(f"avid:{tag}:draw{step}", bs * K, N - 1)).view(bs, K)
            idx = draw + (draw >= y.unsqueeze(1)).long()
            crit.nce_average.sample_negatives = lambda yy, KK, _i=idx: _i
            loss, tb = crit(v, a, y)
            loss.backward()
            out[f"{tag}_y{step}"], out[f"{tag}_idx{step}"] = y.numpy(), idx.numpy()
            out[f"{tag}_loss{step}"] = loss.item()
            for k in tb:
                out[f"{tag}_tb{step}_{k.replace('/', '_')}"] = float(torch.as_tensor(tb[k]).detach())
            out[f"{tag}_gv{step}"], out[f"{tag}_ga{step}"] = v.grad.numpy().copy(), a.grad.numpy().copy()
            out[f"{tag}_Z{step}"] = float(crit.criterion.avg_exp_score)
            out[f"{tag}_v1rows{step}"] = crit.nce_average.view1_mem[y].numpy().copy()
            out[f"{tag}_v2rows{step}"] = crit.nce_average.view2_mem[y].numpy().copy()
        out[f"{tag}_state_keys"] = np.array(sorted(crit.state_dict().keys()))
    np.savez_compressed(os.path.join(args.out, "avid.npz"), **out)

    # ---------------------------------------------------------------- 4/5. CMA sampling + top-K
    out = {}
    N, Pk = 500, 32
    v1, v2 = det_bank("cma:v1", N), det_bank("cma:v2", N)
    for kind in ["consensus", "union", "video", "audio"]:
        smp = CMASampler(v1, v2, {"type": kind, "pos_k": Pk})
        qj, qd = queue.Queue(), queue.Queue()
        smp.sample_dispatcher(qj, workers=1)
        smp.sample_instance(0, qj, qd)
        out[f"topk_{kind}"] = smp.sample_gather(qd, workers=1).astype(np.int32)
    crit = criterions.AVID_CMA.__new__(criterions.AVID_CMA)  # build by hand: ctor needs GPUs for find_correspondences
    torch.nn.Module.__init__(crit)
    from criterions.avid_cma import AVIDSimilarityPositiveExpansion
    K, Kw, bs = 64, 16, 4
    na = AVIDSimilarityPositiveExpansion(memory_size=N, embedding_dim=128, num_negatives=K, num_negatives_within=Kw,
                                         sampling_args={"type": "consensus", "pos_k": Pk}, momentum=0.5)
    na.view1_mem.copy_(v1); na.view2_mem.copy_(v2)
    na.register_buffer("positive_set", T(out["topk_consensus"]).int())
    crit.nce_average = na
    crit.xModalInstCoeff, crit.wModalInstCoeff, crit.xModalPosCoeff, crit.wModalPosCoeff = 0.5, 0.0, 0.0, 0.5
    crit.criterion = NCECriterion(N)
    y = T(detgen.det_indices("cma:y", bs, N))
    rand_idx = T(detgen.det_indices("cma:draw", bs * K, N - Pk)).view(bs, K)
    na.multinomial.draw = lambda n, _r=rand_idx: _r.reshape(-1)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        pos_idx, neg_idx = na.memory_sampling(y)
        out["ms_y"], out["ms_rand"] = y.numpy(), rand_idx.numpy()
        out["ms_pos"], out["ms_neg"] = pos_idx.numpy(), neg_idx.numpy()
        v = T(detgen.det_normalish("cma:v", (bs, 128))).requires_grad_(True)
        a = T(detgen.det_normalish("cma:a", (bs, 128))).requires_grad_(True)
        loss, tb = crit(v, a, y)
    loss.backward()
    out["cma_loss"] = loss.item()
    for k in tb:
        out[f"cma_tb_{k.replace('/', '_')}"] = float(tb[k])
    out["cma_gv"], out["cma_ga"] = v.grad.numpy().copy(), a.grad.numpy().copy()
    out["cma_Z"] = float(crit.criterion.avg_exp_score)
    out["cma_v1rows"], out["cma_v2rows"] = na.view1_mem[y].numpy().copy(), na.view2_mem[y].numpy().copy()
    np.savez_compressed(os.path.join(args.out, "cma.npz"), **out)

    # ---------------------------------------------------------------- 6. blocks (train mode)
    out = {}

    def run_block(tag, blk, x_shape):
        load_det_weights(blk, f"blk:{tag}")
        blk.train()
        x = T(detgen.det_normalish(f"blk:{tag}:x", x_shape)).requires_grad_(True)
        yv = blk(x)
        g = T(detgen.det_uniform(f"blk:{tag}:g", tuple(yv.shape)))
        (yv * g).sum().backward()
        out[f"{tag}_y"] = yv.detach().numpy()
        out[f"{tag}_gx"] = x.grad.numpy()
        for n, p in blk.named_parameters():
            g = p.grad.numpy().reshape(-1)
            out[f"{tag}_g_{n}"] = g[:8192].copy()          # leading slice + full norm keep the fixture small
            out[f"{tag}_gnorm_{n}"] = float(np.linalg.norm(g.astype(np.float64)))
        for n, b in blk.named_buffers():
            out[f"{tag}_buf_{n}"] = b.numpy().copy()

    run_block("r2p1d_64_128_s2", BasicR2P1DBlock(64, 128, stride=(2, 2, 2)), (2, 64, 4, 10, 12))
    run_block("r2p1d_64_64", BasicR2P1DBlock(64, 64), (2, 64, 3, 6, 7))
    run_block("b2d_64_128_s2", Basic2DBlock(64, 128, stride=(2, 2)), (2, 64, 9, 13))
    run_block("b2d_64_64", Basic2DBlock(64, 64), (2, 64, 5, 7))
    np.savez_compressed(os.path.join(args.out, "blocks.npz"), **out)

    # ---------------------------------------------------------------- 7. full av_wrapper, 2 clips
    out = {}
    model = models.av_wrapper("R2Plus1D", {"depth": 18}, "Conv2D", {"depth": 10}, proj_dim=[512, 512, 128])
    load_det_weights(model, "w")
    model.train()
    video = T(detgen.det_normalish("in:video", (2, 3, 8, 112, 112)))
    audio = T(detgen.det_normalish("in:audio", (2, 1, 40, 100)))
    ve, ae = model(video, audio)
    gv = T(detgen.det_uniform("in:gv", (2, 128)))
    ga = T(detgen.det_uniform("in:ga", (2, 128)))
    ((ve * gv).sum() + (ae * ga).sum()).backward()
    out["video_emb"], out["audio_emb"] = ve.detach().numpy(), ae.detach().numpy()
    sd = model.state_dict()
    out["state_keys"] = np.array(list(sd.keys()))
    out["state_shapes"] = np.array([str(tuple(v.shape)) for v in sd.values()])
    grads = dict(model.named_parameters())
    for n in ["video_model.conv1.0.weight", "video_model.conv1.1.weight", "video_model.conv1.1.bias",
              "video_model.conv2x.0.spt_conv1.weight", "video_model.conv2x.1.tmp_conv2.weight",
              "video_model.conv3x.0.res_conv.weight", "video_model.conv3x.0.out_bn.weight",
              "video_model.conv5x.1.out_bn.bias", "audio_model.conv1.0.weight", "audio_model.block1.conv1.weight",
              "audio_model.block4.bn2.weight", "video_proj.projection.0.bias", "video_proj.projection.4.weight",
              "audio_proj.projection.2.weight", "audio_proj.projection.4.bias"]:
        g = grads[n].grad.numpy().reshape(-1)
        out[f"grad:{n}"] = g[:4096].copy()
        out[f"gradnorm:{n}"] = float(np.linalg.norm(g.astype(np.float64)))
    for n in ["video_model.conv1.1.running_mean", "video_model.conv1.1.running_var",
              "video_model.conv4x.1.out_bn.running_var", "audio_model.block3.bn1.running_mean",
              "video_model.conv2x.0.spt_bn1.num_batches_tracked"]:
        out[f"buf:{n}"] = sd[n].numpy().copy()
    with torch.no_grad():
        model.eval()
        e = model.video_model(video, return_embs=True)
        for k, t in e.items():
            out[f"eval_video_{k}_absmean"] = float(t.abs().mean())
        e = model.audio_model(audio, return_embs=True)
        for k, t in e.items():
            out[f"eval_audio_{k}_absmean"] = float(t.abs().mean())
        ve2, ae2 = model(video, audio)
        out["eval_video_emb"], out["eval_audio_emb"] = ve2.numpy(), ae2.numpy()
    np.savez_compressed(os.path.join(args.out, "av_wrapper.npz"), **out)

    for f in sorted(os.listdir(args.out)):
        print(f, os.path.getsize(os.path.join(args.out, f)))


if __name__ == "__main__":
    main()
