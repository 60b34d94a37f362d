#!/usr/bin/env python
"""bench.py — the AVID training step on N x MI355X (one process per GPU, RCCL over xGMI).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one full pass of the hot path over one synthetic batch that is already
resident in HBM: R(2+1)D-18 + Conv2D forward -> AVID criterion (alias draw, bank gather, NCE, EMA
bank update) -> backward (gradient all-reduce overlapped) -> fused flat Adam.  Workload = BASELINE.json
configs[1]/[2]: per-GPU batch 64 of 3x8x112x112 video + 1x40x100 audio, 240k x 128 bank, 1024
negatives, weak scaling (global batch 64*N).

Prints ONE JSON line on rank 0 with `roofline` (dominant kernel, HIP-event timed on its launch stream
inside the timed region) and, at N=1, `cpu_baseline` (the oracle's full step timed on the host cores).
"""
import argparse
import json
import os
import statistics
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
for _p in (REPO, os.path.join(REPO, "avid-cma_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")   # see avid_hip/lib.py: must precede HIP runtime initialisation

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

FWD_GFLOP_PER_CLIP = 6.418      # 2*MAC of every conv + linear at the benchmark shapes
STEP_GFLOP_PER_CLIP = 17.83     # fwd + bwd (3x fwd minus the two never-needed stem input gradients)
R2P1D_FWD_GFLOP_PER_CLIP = 6.159   # the video tower's forward alone
PEAK_F32_MFMA_TFLOPS = 157.3    # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD
PEAK_BF16X3_TFLOPS = 2516.6 / 6   # fp32-equivalent rate of split-bf16 products: six v_mfma_f32_32x32x16_bf16 (16x the fp32 rate) per product
PEAK_HBM_GBS = 8000.0


class ClockSampler:
    """Mean shader clock (GHz) over an interval, read from /sys/class/drm/card*/device/pp_dpm_sclk (the level marked '*')
    by a host thread; None where the node does not exist or does not change."""

    def __init__(self, index):
        import glob
        import threading
        nodes = sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"))
        self.path = None
        try:                    # the card whose PCI address is this device's
            pr = torch.cuda.get_device_properties(index)
            want = "%04x:%02x:%02x." % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
            for n in nodes:
                if want in os.path.realpath(os.path.dirname(n)):
                    self.path = n
        except Exception:       # noqa: BLE001
            pass
        if self.path is None and len(nodes) == 1:
            self.path = nodes[0]
        self.samples, self.run, self.thread = [], False, None
        self._threading = threading

    def _read(self):
        try:
            for line in open(self.path):
                if "*" in line:
                    return float(line.split(":")[1].strip().lower().replace("mhz", "").replace("*", "").strip())
        except Exception:       # noqa: BLE001
            return None
        return None

    def _loop(self):
        while self.run:
            v = self._read()
            if v:
                self.samples.append(v)
            time.sleep(0.002)

    def start(self):
        if self.path is None:
            return
        self.run = True
        self.thread = self._threading.Thread(target=self._loop, daemon=True)
        self.thread.start()

    def stop(self):
        self.run = False
        if self.thread is not None:
            self.thread.join()
        return round(sum(self.samples) / len(self.samples) / 1e3, 3) if self.samples else None


def split_kernel(name):
    """Kernels whose fp32 products are assembled from six bf16 MFMAs in the default build/environment
    (the bias / ReLU epilogues of the heads' linear layers stay on the fp32 instruction inside igemm_pk_kernel<4,1,1,2,0>)."""
    if name.startswith(("stem_fwd3", "stem_wgrad3", "igemm_pk_kernel<", "wino2_kernel", "wino2p_kernel", "tconv64_kernel")):
        return True
    if name.startswith(("wgrad_tab_kernel", "wgrad_group_kernel", "twgrad64_kernel")):
        return os.environ.get("AVID_WGRAD_BF16X3", "1") != "0"
    return False


def mfma_peak(name):
    return PEAK_BF16X3_TFLOPS if split_kernel(name) else PEAK_F32_MFMA_TFLOPS


def csrc_digest():
    """sha256 over the kernel sources (avid-cma_amd/csrc/*.hip, common.h, include/avid_hip.h): what profiles/pmc_traffic.json
    is stamped with when tools/pmc_traffic.py writes it (there is no git on the GPU box)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(REPO, "avid-cma_amd", "csrc", "*.hip"))) + \
        [os.path.join(REPO, "avid-cma_amd", "csrc", "common.h"), os.path.join(REPO, "include", "avid_hip.h")]
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()


def cpu_baseline(budget=25.0):
    """BASELINE config 1 on the host cores with the oracle (a *port*, pinned to the reference by
    tests/test_oracle_golden.py): bs=4, 1000-row bank, K=1024; full step incl. Adam.  The step is timed at 8 / 16 / 32 /
    64 / all threads inside one ~25 s budget (a batch of 4 clips does not scale to 128+ threads: the best count is the
    reference path's best on this host) and the best is reported with its thread count."""
    from oracle import avid_oracle as O
    torch.manual_seed(1234)
    bs, N, K = 4, 1000, 1024
    st = O.OracleStep(num_data=N, num_negatives=K)
    g = torch.Generator().manual_seed(1234)
    video = torch.randn(bs, 3, 8, 112, 112, generator=g)
    audio = torch.randn(bs, 1, 40, 100, generator=g)

    def step():
        y = torch.randperm(N, generator=g)[:bs]
        idx = torch.randint(0, N - 1, (bs, K), generator=g)
        idx = idx + (idx >= y[:, None]).long()
        t0 = time.perf_counter()
        st.forward_backward(video, audio, y, idx)
        t1 = time.perf_counter()
        st.opt.step()
        return time.perf_counter() - t0, t1 - t0

    ncpu = os.cpu_count() or 1
    all_threads = torch.get_num_threads()
    counts = sorted({c for c in (8, 16, 32, 64, all_threads) if c <= max(all_threads, 8)})
    t_start = time.perf_counter()
    step()                                            # warm-up (allocations, first-touch)
    sweep, spent = {}, 0.0
    for c in counts:                                  # two steps per count: the second is the sample
        torch.set_num_threads(c)
        step()
        sweep[c] = step()
        if time.perf_counter() - t_start > 0.6 * budget:
            break
    best = min(sweep, key=lambda c: sweep[c][0])
    torch.set_num_threads(best)
    times, fb = [sweep[best][0]], [sweep[best][1]]
    while time.perf_counter() - t_start < budget and len(times) < 12:
        a, b = step()
        times.append(a)
        fb.append(b)
    torch.set_num_threads(all_threads)
    med = statistics.median(times)
    return {"value": round(bs / med, 3), "unit": "clips/s", "cores": best, "kind": "port",
            "fwd_bwd_nce_only": round(bs / statistics.median(fb), 3),      # without the Adam step
            "thread_sweep_clips_s": {str(c): round(bs / v[0], 3) for c, v in sweep.items()},
            "os_cpu_count": ncpu,
            "sample": f"config1: bs=4 3x8x112x112+1x40x100, bank 1000x128, K=1024, fwd+NCE+bwd+Adam fp32, median of "
                      f"{len(times)} steps at the best thread count ({best} of {sorted(sweep)} tried; "
                      f"{time.perf_counter() - t_start:.1f} s CPU work)"}


def extra_configs(engine, model, video, audio, dev, lib, steps=10, warmup=3):
    """BASELINE configs 4 and 5 on this GPU, after the timed region (same model, same batch; only the criterion
    changes): cfg5 = AVID on the AudioSet-scale 2M x 128 banks (step rate + achieved bank-gather bandwidth of
    bank_scores_fwd from its HIP events), cfg4 = AVID_CMA InstX-N1024-PosW-N64-Top32 on the 240k banks (step rate +
    one whole find_correspondences()).  Not part of `value`."""
    import criterions
    bs = video.shape[0]
    res = {}

    def rate(crit, N):
        engine.criterion = crit
        g = torch.Generator().manual_seed(4321)
        ids = torch.stack([torch.randperm(N, generator=g)[:bs] for _ in range(warmup + steps)]).to(dev)
        for i in range(warmup):
            engine.step(video, audio, ids[i])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            engine.step(video, audio, ids[warmup + i])
        torch.cuda.synchronize()
        return bs * steps / (time.perf_counter() - t0), ids

    # ---- config 5: configs/main/avid/audioset/Cross-N1024.yaml (num_data 1 784 108 -> 2M rows)
    N5 = 2_000_000
    crit5 = criterions.AVID(num_data=N5, embedding_dim=model.out_dim, num_negatives=1024, momentum=0.5,
                            xModal_coeff=1., wModal_coeff=0., device=dev.index)
    clips5, ids5 = rate(crit5, N5)
    from avid_hip import ops
    overlap, model.overlap_towers = model.overlap_towers, False
    defer, ops.DEFER_WGRAD = ops.DEFER_WGRAD, 0     # single stream: an event pair brackets exactly one kernel
    lib.timing_enable(True)
    for i in range(3):
        engine.step(video, audio, ids5[i])
    torch.cuda.synchronize()
    k = lib.timing_report()
    lib.timing_enable(False)
    model.overlap_towers, ops.DEFER_WGRAD = overlap, defer
    # the fused criterion kernel (normalize -> gather both banks -> scores -> NCE -> gradient): its algorithmic bytes
    # are the gathered rows of both banks, READ once (no snapshot is written any more)
    xf = k.get("xmodal_fused_kernel")
    res["cfg5"] = {"bank_rows": N5, "clips_s": round(clips5, 1),
                   "bank_gather_read_GBs": round(xf["bytes"] / (xf["ms"] * 1e-3) / 1e9, 1) if xf else None,
                   "bank_gather_read_bytes_per_step": int(xf["bytes"] / xf["launches"]) if xf else None,
                   "xmodal_fused_us": round(xf["ms"] / xf["launches"] * 1e3, 2) if xf else None}
    del crit5
    torch.cuda.empty_cache()
    # ---- config 4: configs/main/avid-cma/kinetics/InstX-N1024-PosW-N64-Top32.yaml:47-62
    N4 = 240_000
    crit4 = criterions.AVID_CMA(num_data=N4, embedding_dim=model.out_dim, num_negatives=1024, num_negatives_within=64,
                                momentum=0.5, xModalInstCoeff=1., wModalInstCoeff=0., xModalPosCoeff=0.,
                                wModalPosCoeff=1., sampling_args={"type": "consensus", "pos_k": 32}, resample_freq=-1,
                                device=dev.index)          # (the constructor already searched once: warm)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    crit4.nce_average.find_correspondences()
    torch.cuda.synchronize()
    find_s = time.perf_counter() - t0
    clips4, _ = rate(crit4, N4)
    res["cfg4"] = {"bank_rows": N4, "clips_s": round(clips4, 1), "find_correspondences_s": round(find_s, 3),
                   "search_TFLOPs": round(4.0 * N4 * N4 * 128 / find_s / 1e12, 1)}
    return res


def reference_loop(model, crit, video, audio, ids, dev, steps=30, warmup=6, dropin=False):
    """clips/s of the REFERENCE's loop shape on the same kernels (main-avid.py:155-180, utils/main_utils.py:112,250):
    torch DistributedDataParallel around the model (one-rank group), torch.optim.Adam, and the host synchronisation
    ``loss.item()`` between the criterion and ``zero_grad / backward / step`` — what a user of the unmodified driver gets,
    against ``TrainStep`` (flat buffers, fused Adam, gradients written in place, helper streams, no host sync).
    ``dropin``: the SAME loop with the two objects the reference's factories build swapped for this build's
    (utils/main_utils.py:112 -> avid_hip.parallel.DistributedDataParallel, :250 -> avid_hip.parallel.Adam)."""
    if dropin:
        from avid_hip.parallel import DistributedDataParallel, Adam
    else:
        from torch.nn.parallel import DistributedDataParallel
        Adam = torch.optim.Adam
    own_group = not dist.is_initialized()
    if own_group:
        import socket
        sk = socket.socket()
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
        sk.close()
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    overlap, model.overlap_towers = model.overlap_towers, False      # (DDP's reducer knows one stream)
    # like for like: torch's reducer all-reduces its buckets on a one-rank group too, so the drop-in is made to (AVID_FORCE_DIST)
    force = os.environ.get("AVID_FORCE_DIST")
    if dropin:
        os.environ["AVID_FORCE_DIST"] = "1"
    try:
        ddp = DistributedDataParallel(model, device_ids=[dev.index])
        opt = Adam(model.parameters(), lr=2e-4, betas=(0.9, 0.999), weight_decay=1e-5)

        def one(i):
            v, a = ddp(video, audio)
            loss, _ = crit(v, a, ids[i % ids.shape[0]])
            val = loss.item()                         # main-avid.py:174 (meters), before the backward pass
            opt.zero_grad()
            loss.backward()
            opt.step()
            return val
        for i in range(warmup):
            one(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            one(warmup + i)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        del ddp, opt
    finally:
        model.overlap_towers = overlap
        if dropin:
            os.environ.pop("AVID_FORCE_DIST", None)
            if force is not None:
                os.environ["AVID_FORCE_DIST"] = force
        if own_group:
            dist.destroy_process_group()
    return {"clips_s": round(video.shape[0] * steps / dt, 1), "ms_per_step": round(dt / steps * 1e3, 3),
            "what": ("avid_hip.parallel.DistributedDataParallel (1 rank) + avid_hip.parallel.Adam + loss.item() per step: the "
                     "reference's loop with the two objects an UNMODIFIED main-avid.py gets from utils.main_utils when "
                     "avid-cma_amd is ahead of the reference on PYTHONPATH (avid-cma_amd/utils/main_utils.py)") if dropin else
                    "torch DDP (1 rank) + torch.optim.Adam + loss.item() per step, same model / criterion kernels "
                    "(what AVID_DROPIN=0 leaves the reference's factories with)"}


def forward_roofline(model, video, lib, reps=3):
    """The R(2+1)D forward alone (BASELINE.json's target is quoted on it): HIP-event time of its MFMA kernels on one
    stream; direct-form flops (6.159 GFLOP / clip) and the multiply-adds really executed (the Winograd
    layers execute 16 / 36 of theirs)."""
    with torch.no_grad():
        model.video_model(video)
        torch.cuda.synchronize()
        lib.timing_enable(True)
        for _ in range(reps):
            model.video_model(video)
        torch.cuda.synchronize()
        k = lib.timing_report()
        lib.timing_enable(False)
    mf = {n: v for n, v in k.items() if v["flops"] > 0}
    ms = sum(v["ms"] for v in mf.values()) / reps
    executed = sum(v["flops"] for v in mf.values()) / reps
    # the same executed flops against the peak of the instruction each kernel ISSUES (SURVEY 8(d)'s denominator): the time the
    # kernels would need at their own peaks (fp32 MFMA 157.3 TF; split-bf16 products 2516.6 / 6 = 419.4 TF) over the time taken
    ideal_ms = sum(v["flops"] / reps / (mfma_peak(n) * 1e12) * 1e3 for n, v in mf.items())
    direct = R2P1D_FWD_GFLOP_PER_CLIP * 1e9 * video.shape[0]
    all_ms = sum(v["ms"] for v in k.values()) / reps
    return {"frac": round(ideal_ms / ms, 4),          # the forward's headline fraction: of the peak of the instructions issued
            "frac_of_issued_peak": round(ideal_ms / ms, 4),
            "mfma_kernels_ms": round(ms, 3), "all_kernels_ms": round(all_ms, 3),
            "executed": {"gflop_per_clip": round(executed / video.shape[0] / 1e9, 3),
                         "achieved": round(executed / (ms * 1e-3) / 1e12, 2),
                         "frac_of_f32_mfma_peak": round(executed / (ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4)},
            # direct-form flops (2*M*N*K of every layer; Winograd never issues 20/36 of them): a throughput figure for
            # comparisons with other implementations, NOT a roofline fraction — no peak is applied to it
            "direct_form_equivalent": {"gflop_per_clip": R2P1D_FWD_GFLOP_PER_CLIP,
                                       "tflops": round(direct / (ms * 1e-3) / 1e12, 2)},
            "note": "the video tower's forward MFMA kernels (stem, implicit GEMM, Winograd) on one stream.  frac = "
                    "frac_of_issued_peak: the executed flops of every kernel priced at the peak of the matrix instruction "
                    "it issues (fp32 157.3 TF; split-bf16 products 2516.6 / 6 = 419.4 TF fp32-equivalent) over the time "
                    "taken (SURVEY 8(d)'s denominator); executed = the multiply-adds issued (in-image Winograd tiles only)"}


def launcher_command(gpus, argv, visible, env):
    """The command `bench.py --gpus N` re-executes itself through when nobody has started its ranks — what the reference's
    own entry point does with mp.spawn (main-avid.py:69-78: one process per GPU, started by the script the user ran) —
    or None when this process IS a rank (WORLD_SIZE set by torch.distributed.run / the driver) or a plain one-GPU run.
    `visible` = GPUs this host shows; raises SystemExit naming that count when N exceeds it.  AVID_FORCE_DIST=1 sends a
    one-GPU run the same way (a one-rank RCCL group: the 1-GPU box check of the N > 1 path)."""
    if "WORLD_SIZE" in env:
        return None
    forced = env.get("AVID_FORCE_DIST", "0") == "1"
    if gpus <= 1 and not forced:
        return None
    if gpus < 1:
        raise SystemExit(f"bench.py: --gpus {gpus}: need at least one GPU")
    if gpus > visible:
        raise SystemExit(f"bench.py: --gpus {gpus} but this host shows {visible} GPU{'s' if visible != 1 else ''} "
                         f"(torch.cuda.device_count() = {visible}): one process per GPU, one GPU per process")
    port = env.get("MASTER_PORT")
    if port is None:
        import socket
        with socket.socket() as sk:          # a free port for the rendezvous
            sk.bind(("127.0.0.1", 0))
            port = str(sk.getsockname()[1])
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *argv]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=64, help="per-GPU batch (BASELINE configs[1]/[2]: 64)")
    ap.add_argument("--bank", type=int, default=240000)
    ap.add_argument("--negatives", type=int, default=1024)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the config 4 / config 5 side measurements")
    ap.add_argument("--graph", type=int, default=-1,
                    help="replay the whole step as one hipGraph (1/0; default 0: the eager path overlaps the two "
                         "towers on two streams and is GPU-bound — host issue ~10 ms vs ~20 ms of kernels)")
    ap.add_argument("--breakdown", action="store_true", help="also print the per-kernel HIP-event table (stderr)")
    ap.add_argument("--cu-reserve", type=int, default=-1,
                    help="CUs the persistent kernels leave to RCCL's workgroups when gradients are all-reduced (N > 1); "
                         "default: 0 and 8 are both timed during warm-up and the faster is kept")
    args = ap.parse_args()

    cmd = launcher_command(args.gpus, sys.argv[1:], torch.cuda.device_count(), os.environ)
    if cmd is not None:
        # rank 0 of the child job prints the one JSON line on the stdout it inherits; this process only waits
        import subprocess
        sys.stdout.flush()
        raise SystemExit(subprocess.call(cmd))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 or os.environ.get("AVID_FORCE_DIST", "0") == "1":
        # (before the HIP runtime starts) one hardware queue per stream — compute, audio tower, bucket launches, RCCL's
        # own: with the runtime's default of four, two of them share a queue and wait behind each other (one-rank
        # RCCL group: 4320 -> 4560 clips/s)
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    if local_rank >= torch.cuda.device_count():
        raise SystemExit(f"bench.py: rank {rank} wants GPU {local_rank} but this host shows {torch.cuda.device_count()}")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # AVID_FORCE_DIST=1 drives the multi-GPU code path (RCCL group, bucketed all-reduce, bank all-gather, barrier
    # + max-over-ranks timing) on a single-rank group: the 1-GPU box check of what the driver runs at N = 2/4/8
    use_dist = world > 1 or os.environ.get("AVID_FORCE_DIST", "0") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import models
    import criterions
    from avid_hip import lib
    from avid_hip.parallel import TrainStep

    torch.manual_seed(0)
    model = models.av_wrapper("R2Plus1D", {"depth": 18}, "Conv2D", {"depth": 10}, proj_dim=[512, 512, 128]).to(dev)
    model.train()
    crit = criterions.AVID(num_data=args.bank, embedding_dim=model.out_dim, num_negatives=args.negatives,
                           momentum=0.5, xModal_coeff=1., wModal_coeff=0., device=local_rank)
    engine = TrainStep(model, crit, lr=2e-4, weight_decay=1e-5)

    bs = args.batch
    g = torch.Generator().manual_seed(1234 + rank)
    video = torch.randn(bs, 3, 8, 112, 112, generator=g).to(dev)
    audio = torch.randn(bs, 1, 40, 100, generator=g).to(dev)
    total = args.warmup + args.steps
    gp = torch.Generator().manual_seed(99)                     # same permutation on every rank, sharded by rank
    ids = torch.stack([torch.randperm(args.bank, generator=gp)[:bs * world] for _ in range(total)])
    ids = ids[:, rank * bs:(rank + 1) * bs].contiguous().to(dev)

    def sync():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    use_graph = False if args.graph < 0 else bool(args.graph)
    from avid_hip import ops as _ops
    cu_info = None
    if args.cu_reserve >= 0:
        cu_info = {"cu_reserve": args.cu_reserve, "cus_planned": _ops.set_cu_budget(_ops.cu_budget() - args.cu_reserve if args.cu_reserve else 0)}
    for i in range(args.warmup):
        engine.step(video, audio, ids[i])
    sync()
    if (world > 1 or os.environ.get("AVID_BENCH_CU_AB", "0") == "1") and use_dist and args.cu_reserve < 0:
        # RCCL's workgroups share the CUs with the persistent kernels, whose tile deal assumes every slot:
        # time three steps with every CU planned and three with one CU per XCD left free, keep the faster (max over ranks)
        def timed3():
            sync()
            t = time.perf_counter()
            for i in range(3):
                engine.step(video, audio, ids[i % total])
            sync()
            tt = torch.tensor([(time.perf_counter() - t) / 3 * 1e3], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            return float(tt.item())
        all_cus = _ops.cu_budget()
        ms_all = timed3()
        _ops.set_cu_budget(all_cus - 8)
        for i in range(2):                     # (the launch programs are recompiled for the new plans: untimed)
            engine.step(video, audio, ids[i])
        ms_res = timed3()
        keep = 8 if ms_res < 0.995 * ms_all else 0
        planned = _ops.set_cu_budget(all_cus - keep if keep else 0)
        for i in range(2):
            engine.step(video, audio, ids[i])
        sync()
        cu_info = {"cu_reserve": keep, "cus_planned": planned, "cu_reserve_ab_ms": {"0": round(ms_all, 3), "8": round(ms_res, 3)}}
    # ---- the distributed arrangement is ONE: weight gradients on the trailing stream, every gradient bucket all-reduced
    # on the collectives' stream as soon as it is complete (avid_hip/parallel.py GradBuckets; the four streams placed on
    # four dispatch pipes by avid_hip/streams.py — the three-way A/B of round 3 priced a pipe collision)
    dist_info = None
    if use_dist:
        dist_info = {"dist_mode": "trailing weight gradients + buckets reduced as they complete"}
        if args.graph < 0 and os.environ.get("AVID_BENCH_GRAPH_AB", "0") == "1":
            try:
                def timed(fn, n=3):
                    sync()
                    t = time.perf_counter()
                    for i in range(n):
                        fn(i)
                    sync()
                    tt = torch.tensor([(time.perf_counter() - t) / n * 1e3], dtype=torch.float64, device=dev)
                    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                    return float(tt.item())
                eager_ms = timed(lambda i: engine.step(video, audio, ids[i % total]))
                engine.capture(video, audio, ids[0])
                engine.replay(index=ids[0])
                graph_ms = timed(lambda i: engine.replay(index=ids[i % total]))
                dist_info["graph_ab_ms"] = {"eager": round(eager_ms, 3), "graph": round(graph_ms, 3)}
                if graph_ms < 0.99 * eager_ms:
                    use_graph = True
                    dist_info["dist_mode"] = "graph"
            except Exception as e:                          # noqa: BLE001
                dist_info["graph_ab_error"] = repr(e)[:200]
        engine.buckets.measure = True
    # Per-kernel HIP-event pass (events on the launch stream, library-side): a few eager steps of the same
    # workload OUTSIDE the timed region, so the instrumentation does not perturb `value`.
    # The pass runs single-stream (tower overlap off) so an event pair brackets exactly one kernel.
    from avid_hip import ops
    overlap, defer = model.overlap_towers, ops.DEFER_WGRAD
    model.overlap_towers = False
    ops.DEFER_WGRAD = 0                    # (the weight gradients otherwise trail on helper streams, next to other kernels)
    lib.timing_enable(True)
    kern_steps = min(3, args.steps)
    for i in range(kern_steps):
        engine.step(video, audio, ids[args.warmup + i])
    torch.cuda.synchronize()
    kern = lib.timing_report()
    lib.timing_enable(False)
    model.overlap_towers, ops.DEFER_WGRAD = overlap, defer
    if use_graph and engine.graph is None:
        engine.capture(video, audio, ids[0])
        engine.replay(index=ids[0])
    sync()
    # the shader clock DURING the timed region, sampled by a host thread from the driver's sysfs node every 2 ms (a probe
    # wave spinning beside the steps was tried first: it takes one CU's registers, and a persistent kernel that cannot place
    # one of its workgroups pays a whole extra round — 11.3 -> 12.0 ms per step)
    sampler = ClockSampler(local_rank)
    sampler.start()
    t0 = time.perf_counter()
    loss = None
    for i in range(args.steps):
        if use_graph:
            loss = engine.replay(index=ids[args.warmup + i])
        else:
            loss = engine.step(video, audio, ids[args.warmup + i])
    sync()
    dt = time.perf_counter() - t0
    clock_ghz = sampler.stop()
    # host issue time: how long the host needs to ENQUEUE a step (5 steps issued back to back, nothing waited for) —
    # the margin between it and ms_per_step is what a slower host may eat before the GPU starves
    t0 = time.perf_counter()
    for i in range(min(5, args.steps)):
        if use_graph:
            engine.replay(index=ids[args.warmup + i])
        else:
            engine.step(video, audio, ids[args.warmup + i])
    host_issue_ms = (time.perf_counter() - t0) / min(5, args.steps) * 1e3
    sync()
    if use_dist:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
        # per-rank facts for the record: device, host issue time, exposed wait for the gradient collectives
        exposed = engine.buckets.exposed_wait_ms()
        engine.buckets.measure = False
        mine = torch.tensor([float(torch.cuda.current_device()), host_issue_ms, -1.0 if exposed is None else exposed],
                            dtype=torch.float64, device=dev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        dist_info.update({"rccl_world": dist.get_world_size(), "backend": dist.get_backend(),
                          "devices": [int(t[0]) for t in allr],
                          "host_issue_ms_per_rank": [round(float(t[1]), 3) for t in allr],
                          "allreduce_exposed_ms_per_rank": [round(float(t[2]), 3) for t in allr],
                          "bucket_count": len(engine.buckets.bounds),
                          "gradient_bytes": int(engine.flat.numel * 4)})
    loss_val = float(loss)
    from avid_hip import streams as _streams
    stream_report = _streams.report(dev)

    if rank == 0:
        ms = dt / args.steps * 1e3
        clips = bs * world * args.steps / dt
        # every MFMA kernel of the step: implicit-GEMM forward / dgrad, weight gradients, the two LDS-patch stems
        mfma = {k: v for k, v in kern.items()
                if v["flops"] > 0 and ("igemm" in k or "wgrad" in k or k.startswith(("stem_", "wino_", "wino2_", "wino2p_", "tconv")))}
        dom = max(mfma, key=lambda k: mfma[k]["ms"])
        d = mfma[dom]
        ach = d["flops"] / (d["ms"] * 1e-3) / 1e12
        # HBM traffic of the dominant kernel from the committed PMC passes (tools/pmc.sh + tools/pmc_traffic.py:
        # FETCH_SIZE doubled per MI355X_MICROARCH.md's gfx950 note, + WRITE_SIZE), bytes per launch
        tf = os.path.join(REPO, "profiles", "pmc_traffic.json")
        pmc = json.load(open(tf)) if os.path.exists(tf) else {}
        traffic_source = pmc.pop("_source", None)
        now = csrc_digest()
        if traffic_source is None or traffic_source.get("csrc_sha256") != now:
            # the table was taken from other kernel sources than the ones running now: no stale figure in the record
            print("bench.py: profiles/pmc_traffic.json was collected from different kernel sources "
                  f"({(traffic_source or {}).get('csrc_sha256', 'unstamped')[:12]} vs {now[:12]} now): roofline.traffic is null; "
                  "re-run tools/collect_profiles.sh on the GPU box and commit the regenerated table", file=sys.stderr)
            traffic_source = dict(traffic_source or {}, stale=True, csrc_sha256_now=now)
            pmc = {}
        traffic = pmc.get(dom)
        if traffic is None and pmc:     # a renamed / new dominant kernel must not report a stale figure: null + a loud note
            print(f"bench.py: profiles/pmc_traffic.json has no entry for the dominant kernel {dom!r}: roofline.traffic "
                  f"is null; re-run tools/collect_profiles.sh on the GPU box and commit the regenerated file",
                  file=sys.stderr)
        conv_ms = sum(v["ms"] for v in mfma.values()) / kern_steps
        conv_tf = sum(v["flops"] for v in mfma.values()) / (sum(v["ms"] for v in mfma.values()) * 1e-3) / 1e12
        out = {
            "metric": "clips/sec (video+audio fwd+bwd+NCE+Adam)",
            "value": round(clips, 2), "unit": "clips/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None,
            # fp32 storage, accumulation and tolerances end to end; the stem, weight-gradient and 64x64-wave-tile convolution
            # kernels assemble each fp32 product from six bf16 MFMAs (operands split into three bf16 terms in registers:
            # error vs fp64 at or below the fp32 MFMA instruction's)
            "dtype": "f32 (bf16x3 MFMA)",
            "data": "synthetic",
            "config": {"workload": "AVID Cross-N1024 step: R(2+1)D-18 + Conv2D-10 + heads [512,512,128], "
                                   "3x8x112x112 video + 1x40x100 audio",
                       "per_gpu_batch": bs, "global_batch": bs * world, "bank_rows": args.bank,
                       "negatives": args.negatives, "parallelism": f"dp{world}", "optimizer": "adam(2e-4, wd 1e-5)",
                       "hipgraph": use_graph, "host_issue_ms_per_step": round(host_issue_ms, 3),
                       "stream_placement": stream_report,
                       "loss": round(loss_val, 5), **(dist_info or {}), **(cu_info or {})},
            "roofline": {"bound": "mfma", "kernel": dom, "achieved": round(ach, 2), "peak": round(mfma_peak(dom), 1),
                         "peak_basis": ("fp32-equivalent: v_mfma_f32_32x32x16_bf16 dense peak / 6 (six bf16 MFMAs per fp32 product)"
                                        if split_kernel(dom) else "v_mfma_f32_32x32x2_f32"),
                         "unit": "TFLOP/s", "frac": round(ach / mfma_peak(dom), 4), "traffic": traffic,
                         "traffic_source": dict({"file": "profiles/pmc_traffic.json"}, **(traffic_source or {})),
                         "shader_clock_ghz": clock_ghz,
                         # ms_per_step x mean shader clock: the step in shader cycles.  Boxes of this pool hold 2.11 - 2.29 GHz under
                         # this load and the step time follows the clock: this figure compares runs across boxes
                         "mcycles_per_step": None if clock_ghz is None else round(ms * clock_ghz, 2),
                         "algorithmic_bytes_per_launch": round(d["bytes"] / d["launches"]),
                         "launches_per_step": d["launches"] / kern_steps,
                         "avg_launch_ms": round(d["ms"] / d["launches"], 4),
                         # every MFMA kernel of the step (the dominant one is whichever has the most time: the forward
                         # `...,0>` and the input-gradient `...,1>` instantiations of the 128x64 tile are within 2 % of
                         # each other, the latter also carries the fused BatchNorm-backward sums)
                         "flop_accounting": "achieved / frac of every kernel = multiply-adds the kernel EXECUTES; the "
                                            "Winograd kernels (wino_*, wino2_*) execute 16/36 of the direct form's: their "
                                            "direct_equivalent figure prices the same launches at 2*M*N*K; "
                                            "step_algorithmic and r2p1d_forward.direct_form_equivalent are direct-form",
                         "mfma_kernels": {k: dict({"ms_per_step": round(v["ms"] / kern_steps, 3),
                                                   "achieved": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2),
                                                   "peak": round(mfma_peak(k), 1),
                                                   "frac": round(v["flops"] / (v["ms"] * 1e-3) / 1e12 / mfma_peak(k), 4),
                                                   "flops": "fp32-equivalent (six bf16 MFMAs per product, bf16x3)" if split_kernel(k) else "executed"},
                                                  **({"direct_equivalent": round(2.25 * v["flops"] / (v["ms"] * 1e-3) / 1e12, 2)}
                                                     if k.startswith(("wino_", "wino2_", "wino2p_")) else {}),
                                                  # counter HBM traffic per launch (profiles/pmc_traffic.json) next to
                                                  # the algorithmic bytes of the same launches
                                                  **({"traffic": pmc[k], "algorithmic_bytes_per_launch": round(v["bytes"] / v["launches"])}
                                                     if k in pmc else {}))
                                          for k, v in sorted(mfma.items(), key=lambda kv: -kv[1]["ms"])},
                         "all_conv_kernels": {"ms_per_step": round(conv_ms, 3), "achieved": round(conv_tf, 2),
                                              "frac": round(conv_tf / PEAK_F32_MFMA_TFLOPS, 4),
                                              "frac_of": "v_mfma_f32_32x32x2_f32 peak 157.3 TF — the common yardstick; the "
                                                         "split-bf16 kernels are priced against their own peak above"},
                         "step_algorithmic": {"gflop_per_clip": STEP_GFLOP_PER_CLIP,
                                              "achieved": round(clips / world * STEP_GFLOP_PER_CLIP / 1e3, 2),
                                              "frac": round(clips / world * STEP_GFLOP_PER_CLIP / 1e3
                                                            / PEAK_F32_MFMA_TFLOPS, 4)}},
        }
        if args.breakdown:
            tot = sum(v["ms"] for v in kern.values())
            print(f"{'kernel':34s} {'launch/step':>11s} {'ms/step':>9s} {'%':>6s} {'TFLOP/s':>9s} {'GB/s(alg)':>10s}",
                  file=sys.stderr)
            for k, v in sorted(kern.items(), key=lambda kv: -kv[1]["ms"]):
                tf = v["flops"] / (v["ms"] * 1e-3) / 1e12 if v["flops"] else 0.0
                gb = v["bytes"] / (v["ms"] * 1e-3) / 1e9
                print(f"{k:34s} {v['launches'] / kern_steps:11.1f} {v['ms'] / kern_steps:9.3f} "
                      f"{100 * v['ms'] / tot:6.1f} {tf:9.2f} {gb:10.1f}", file=sys.stderr)
            print(f"timed kernels {tot / kern_steps:.3f} ms/step of {ms:.3f} ms wall", file=sys.stderr)
        # (side measurements must never cost the line: a failure is reported inside it)
        try:
            out["roofline"]["r2p1d_forward"] = forward_roofline(model, video, lib)
        except Exception as e:                              # noqa: BLE001
            out["roofline"]["r2p1d_forward"] = {"error": repr(e)[:300]}
        if world == 1 and not args.no_extra:
            try:
                out["extra"] = extra_configs(engine, model, video, audio, dev, lib)
            except Exception as e:                          # noqa: BLE001
                out["extra"] = {"error": repr(e)[:300]}
            engine.criterion = crit
            try:
                ref = reference_loop(model, crit, video, audio, ids, dev)
                ref["vs_trainstep"] = round(ref["clips_s"] / clips, 3)
            except Exception as e:                          # noqa: BLE001
                ref = {"error": repr(e)[:300]}
            out["extra"]["reference_loop"] = ref
            # (last: the wrapper re-seats the parameters in flat buffers of its own — `engine` is not used after this)
            try:
                ref = reference_loop(model, crit, video, audio, ids, dev, dropin=True)
                ref["vs_trainstep"] = round(ref["clips_s"] / clips, 3)
            except Exception as e:                          # noqa: BLE001
                ref = {"error": repr(e)[:300]}
            out["extra"]["reference_loop_dropin"] = ref
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline()
            except Exception as e:                          # noqa: BLE001
                out["cpu_baseline"] = {"error": repr(e)[:300]}
        # RCCL prints a version banner through C stdio when a communicator is first created: flush it BEFORE the JSON
        # line so that the line is the last thing on stdout
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
