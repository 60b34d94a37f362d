"""DESIGN.md's table "what runs in a default batch-64 step": every kernel of the newest committed profile set
(profiles/r*_hip_event_breakdown.txt: launches and ms per step on one stream, + the traffic table) with the line of its
definition in avid-cma_amd/csrc/ and the layers it serves.  python tools/kernel_inventory.py > table.md"""
import glob, json, os, re, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROLE = {
    "wino_wgrad_kernel": "weight gradient of the stride-1 3x3 layers (conv2x x4, conv3x x3, conv4x x3, audio block 1) through the Winograd identity; fp32 MFMA",
    "igemm_pk_kernel<4,1,1,2,1>": "input gradients on the 128 x 64 tile, weights pre-split (conv2x temporal with the BatchNorm-backward sums, conv4x / conv5x / audio / heads)",
    "igemm_pk_kernel<4,1,1,2,0>": "forward on the 128 x 64 tile (conv4x / conv5x temporal + strided, audio blocks, heads; K-split tails weight-stationary)",
    "igemm_pk_kernel<4,1,1,4,0>": "forward on the 128 x 128 tile as four waves of 32 x 128, weights pre-split (conv3x temporal / strided / residual, conv4x strided, audio blocks 3-4)",
    "igemm_pk_kernel<4,1,1,4,1>": "input gradients on the same tile (conv3x temporal, audio blocks 3-4)",
    "igemm_pk_kernel<4,1,1,2,1>s2": "strided input gradients (stride-parity classes) of the nine stage-transition layers, weights pre-split",
    "wino2p_kernel<1>": "Winograd F(2x2,3x3) forward of conv2x / conv3x spatial layers (+BatchNorm partial sums); V split once into bf16 planes",
    "wino2p_kernel<2>": "... input gradient with the residual addend (first block's spt_conv1)",
    "wino2p_kernel<4>": "... input gradient + BatchNorm-backward sums",
    "wino2p_kernel<6>": "... input gradient + addend + BatchNorm-backward sums",
    "wino_kernel<1>": "Winograd forward of conv4x spatial and audio block 1 (fp32 MFMA, two workgroups per CU)",
    "wino_kernel<4>": "... input gradient + BatchNorm-backward sums",
    "wino_kernel<6>": "... + addend",
    "wgrad_group_kernel": "weight gradients of the small layers, up to 12 per launch (conv3x-5x temporal / strided / residual, audio, heads)",
    "wgrad_tab_kernel<1,3>": "weight gradient of the audio stem-sized 64-channel strided layer",
    "stem_fwd3p_kernel<3,3>": "video stem forward (3,7,7)/s(1,2,2), patch split once into bf16 planes (+BatchNorm partial sums)",
    "stem_wgrad3_kernel<3,3>": "video stem weight gradient",
    "stem_fwd_kernel<1,1>": "audio stem forward 7x7/s2", "stem_wgrad_kernel<1,1>": "audio stem weight gradient",
    "tconv64_kernel<0>": "conv2x (3,1,1) forward: rows staged once for three taps, weights resident in LDS; applies the BatchNorm (+ReLU) in front of it",
    "twgrad64_kernel": "conv2x (3,1,1) weight gradient, taps share split fragments; applies the same BatchNorm to x",
    "bn_apply_kernel": "BatchNorm (+ReLU) apply of the large layers whose consumer cannot apply it (conv2x spatial consumers, out_bn)",
    "bn_bwd_apply_kernel": "BatchNorm backward apply, large layers", "bn_fin_apply_kernel": "finalize + apply in one launch, small layers (M*C <= 8 M)",
    "bn_bwd_fin_apply_kernel": "backward finalize + apply, small layers", "bn_pool_fwd_kernel": "video stem tail: BatchNorm + ReLU + MaxPool(1,3,3) in one pass",
    "bn_pool_bwd_apply_kernel": "... backward apply", "bn_pool_bwd_partial_kernel": "... backward sums", "bn_bwd_partial_kernel": "BatchNorm backward sums of the two layers that feed the global pools",
    "weight_transpose_batched_kernel": "once per step: every weight repacked / Winograd-transformed / pre-split for its kernels (one launch over a table)",
    "wino_weight_kernel": "forward Winograd transforms of the weights (per layer)", "wgrad_reduce_kernel": "fixed-order sum of weight-gradient slabs",
    "wgrad_group_reduce_kernel": "... of a grouped launch", "splitk_reduce_kernel": "fixed-order sum of split-K slabs", "splitk_reduce_stats_kernel": "... + BatchNorm partial sums",
    "splitk_reduce_bnb_kernel": "... + BatchNorm-backward sums", "adam_flat_kernel": "Adam over the flat parameter / gradient / moment buffers, one launch",
    "xmodal_fused_kernel": "criterion: draw, bank gather, scores, NCE forward + backward + bank update staging in one kernel",
}
def main():
    bd = sorted(glob.glob(os.path.join(REPO, "profiles", "r*_hip_event_breakdown.txt")))[-1]
    tag = re.match(r"(r\d+_\w)_", os.path.basename(bd)).group(1)
    traffic = json.load(open(os.path.join(REPO, "profiles", "pmc_traffic.json")))
    line = json.loads(open(os.path.join(REPO, "profiles", tag + "_bench.json")).read().strip().splitlines()[-1])
    mk = line["roofline"]["mfma_kernels"]
    defs = {}
    for f in sorted(glob.glob(os.path.join(REPO, "avid-cma_amd", "csrc", "*.hip"))):
        for n, l in enumerate(open(f), 1):
            m = re.match(r"__global__.*?void (\w+)\(", l)
            if m: defs.setdefault(m.group(1), f"{os.path.basename(f)}:{n}")
    print(f"| kernel (timer name = rocprofv3 name) | defined | launches / step | ms / step | of its peak | HBM-side bytes : algorithmic | serves |")
    print("|---|---|---|---|---|---|---|")
    for l in open(bd):
        m = re.match(r"(\S+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s*$", l)
        if not m: continue
        k = m.group(1); base = k.split("<")[0]
        fr = f"{mk[k]['frac']:.2f}" if k in mk else "—"
        tr = f"{mk[k]['traffic'] / mk[k]['algorithmic_bytes_per_launch']:.2f}" if k in mk and mk[k].get("traffic") else "—"
        print(f"| `{k}` | `{defs.get(base, '?')}` | {float(m.group(2)):g} | {m.group(3)} | {fr} | {tr} | {ROLE.get(k, '')} |")
    print(f"\n(`profiles/{tag}_*`: {line['value']:.0f} clips/s, {line['ms_per_step']} ms per step at {line['roofline']['shader_clock_ghz']} GHz = "
          f"{line['roofline']['mcycles_per_step']} Mcycles; kernels back to back on one stream; 'of its peak' = executed flops over the peak of the "
          f"matrix instruction the kernel issues: 157.3 TF fp32, 419.4 TF fp32-equivalent for six-bf16-product kernels)")
main()
