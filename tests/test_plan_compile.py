"""Host logic of the launch-program compiler (avid_hip/plan.py), no GPU: the forward / backward programs of the stock
two-tower model compile, every tensor reference stays inside its arena, every parameter gets exactly one gradient
record, and the gradient-buffer layout is the step engine's (parallel.FlatParams)."""
import ctypes as C

import pytest
import torch


@pytest.fixture(scope="module")
def compiled():
    import models
    from avid_hip import plan
    m = models.av_wrapper("R2Plus1D", {"depth": 18}, "Conv2D", {"depth": 10}, proj_dim=[512, 512, 128]).train()
    pl = plan.Plan(m, (4, 3, 8, 112, 112), (4, 1, 40, 100), torch.device("cpu"), True, True, True)
    return m, pl


def test_record_layout_matches_the_library():
    from avid_hip import lib
    assert lib.raw("avid_program_instr_bytes")() == C.sizeof(lib.Instr)


def test_programs_compile_and_references_stay_in_bounds(compiled):
    from avid_hip import plan
    m, pl = compiled
    # 2 stems + 8 blocks x (4 conv + 4 BN) + 3 residual convs + 9 audio layers x 2 + 2 pools + 6 linears + 7 plumbing records (each tower waits once for the weight tables)
    assert pl.n_fwd == 102
    size = {plan.S_FWD: pl.fa_bytes, plan.S_BWD: pl.ba_bytes, plan.S_GRAD: 4 * pl.gnumel, plan.S_AUX: pl.aux_bytes}
    for prog, n in ((pl.fwd_prog, pl.n_fwd), (pl.bwd_prog, pl.n_bwd)):
        for k in range(n):
            r = prog[k]
            assert 0 <= r.op <= 17 and 0 <= r.stream < 4
            for j in range(plan.NREF):
                s, off = r.t[j].slot, r.t[j].off
                assert -1 <= s < pl.n_slots
                if s in size:
                    assert 0 <= off < size[s], (k, j, s, off)
                elif s >= 0:
                    assert off == 0
    # a grouped launch is followed by exactly its items
    k = 0
    while k < pl.n_bwd:
        r = pl.bwd_prog[k]
        if r.op == plan.OP_WGRAD_GROUP:
            n = r.i[0]
            assert 1 <= n <= 12 and all(pl.bwd_prog[k + 1 + j].op == plan.OP_WGRAD_ITEM for j in range(n))
            k += n
        k += 1


def test_every_parameter_gets_one_gradient(compiled):
    m, pl = compiled
    seen = []
    for _, _, ps in pl.grad_ready:
        seen += ps
    assert sorted(seen) == list(range(len(pl.params))) == list(range(sum(1 for _ in m.parameters())))


def test_gradient_layout_is_flatparams(compiled):
    from avid_hip.parallel import FlatParams
    import models
    m2 = models.av_wrapper("R2Plus1D", {"depth": 18}, "Conv2D", {"depth": 10}, proj_dim=[512, 512, 128])
    flat = FlatParams(m2)
    _, pl = compiled
    assert list(flat.offsets) == list(pl.goff) and flat.numel == pl.gnumel


def test_segments_cover_the_program_and_follow_the_buckets(compiled):
    m, pl = compiled
    n = len(pl.params)
    bucket_of = [min(i // 30, 4) for i in range(n)]
    counts = [bucket_of.count(b) for b in range(5)]
    segs = pl.segments(bucket_of, counts)
    ends = [e for e, _ in segs]
    assert ends == sorted(ends) and ends[-1] == pl.n_bwd
    assert sorted(i for _, r in segs for i, _ in r) == list(range(n))
    left = list(counts)
    for e, r in segs[:-1]:
        for i, _ in r:
            left[bucket_of[i]] -= 1
        assert 0 in left                                    # every cut completes a bucket


def test_unknown_module_falls_back(compiled):
    import models
    from avid_hip import plan
    m = models.av_wrapper("R2Plus1D", {"depth": 18}, "Conv2D", {"depth": 10}, proj_dim=[512, 512, 128]).train()
    m.video_model.conv2x[0] = torch.nn.Identity()
    with pytest.raises(plan.Unsupported):
        plan.Plan(m, (2, 3, 8, 64, 64), (2, 1, 40, 100), torch.device("cpu"), True, True, True)


def test_weight_tables_and_the_tail_of_the_backward(compiled):
    """The per-step weight-transform table holds, for every weight, the forms its consumers read — plain transposes
    (mode 0), Winograd transforms for the forward (1 / 3) and the input gradient (2 / 4), bf16-split weights for the forward
    (5) and the input gradient (6) — and no plain transpose for a layer whose input gradient reads the split form.  The
    backward program makes the fourth stream wait for the three others right in front of the video stem's backward: the
    optimizer may update everything but the stem's three parameters from there (``adam_early``)."""
    from avid_hip import plan
    m, pl = compiled
    modes = {}
    for w, off, cout, taps, cin, mode in pl.wt_recs:
        modes.setdefault(id(w), set()).add(mode)
        assert 0 <= off < pl.table_off
    all_modes = set().union(*modes.values())
    assert {0, 5, 6} <= all_modes and all_modes <= {0, 1, 2, 3, 4, 5, 6}
    assert not any({0, 6} <= v and not (v & {2, 4}) for v in modes.values()), "a transposed copy nobody reads"
    stem = m.video_model.conv1
    tail = [pl.goff[[id(p) for p in pl.params].index(id(p))] for p in (stem[0].weight, stem[1].weight, stem[1].bias)]
    assert pl.adam_early == min(tail) and max(tail) < pl.gnumel
    waits = [k for k in range(pl.n_bwd) if pl.bwd_prog[k].op == plan.OP_WAIT and pl.bwd_prog[k].i[0] == plan.ST_COMM]
    assert sorted(pl.bwd_prog[k].i[1] for k in waits) == [plan.ST_MAIN, plan.ST_AUDIO, plan.ST_TRAIL]
    first_stem = min(k for k in range(pl.n_bwd) if pl.bwd_prog[k].op == plan.OP_BN_POOL_BWD)
    assert max(waits) < first_stem


def test_conv2x_temporal_layers_apply_their_inputs_batchnorm_at_batch_64():
    """At the benchmark's batch conv2x's (3,1,1) layers run on tconv64_kernel / twgrad64_kernel, which apply the BatchNorm (+ReLU) in
    front of them while they stage their input (avid_conv_fwd_in / avid_conv_wgrad_in): spt_bn1 and spt_bn2 of both conv2x blocks
    then make statistics only (no output tensor), the four temporal convolutions read the BatchNorm's INPUT + its saved vectors
    in the forward and in the weight gradient — and the forward arena loses the four normalised tensors (103 MB each)."""
    import models
    from avid_hip import plan
    m = models.av_wrapper("R2Plus1D", {"depth": 18}, "Conv2D", {"depth": 10}, proj_dim=[512, 512, 128]).train()
    pl = plan.Plan(m, (64, 3, 8, 112, 112), (64, 1, 40, 100), torch.device("cpu"), True, True, True)
    fwd = [pl.fwd_prog[k] for k in range(pl.n_fwd)]
    bwd = [pl.bwd_prog[k] for k in range(pl.n_bwd)]
    stats_only = [r for r in fwd if r.op == plan.OP_BN_FWD and r.t[5].slot < 0]
    fused_fwd = [r for r in fwd if r.op == plan.OP_CONV_FWD and r.i[1] != 0]
    fused_wgrad = [r for r in bwd if r.op == plan.OP_CONV_WGRAD and r.i[0] != 0]
    assert len(stats_only) == len(fused_fwd) == len(fused_wgrad) == 4
    for r in fused_fwd + fused_wgrad:
        assert (r.d.kt, r.d.kh, r.d.kw, r.d.Cin, r.d.Cout) == (3, 1, 1, 64, 64)
    for r in fused_fwd:
        assert r.i[1] == 2 and r.i[2] == 64 and r.t[7].slot == plan.S_FWD          # ReLU'd BatchNorm of 64 channels, vectors in the arena
        # the convolution reads what the statistics-only BatchNorm normalises
        assert any(b.t[0].slot == r.t[0].slot and b.t[0].off == r.t[0].off and b.t[6].off == r.t[7].off for b in stats_only)
    for r in fused_wgrad:
        assert r.i[0] == 2 and r.i[1] == 64 and r.t[3].slot == plan.S_FWD
    # with the fusion off every BatchNorm writes its output: four more tensors of 64 x 8 x 28 x 28 x 64 floats in the forward arena
    import os
    import subprocess
    import sys
    code = ("import sys, torch; sys.path.insert(0, %r); sys.path.insert(0, %r); import models; from avid_hip import plan; "
            "m = models.av_wrapper('R2Plus1D', {'depth': 18}, 'Conv2D', {'depth': 10}, proj_dim=[512, 512, 128]).train(); "
            "pl = plan.Plan(m, (64, 3, 8, 112, 112), (64, 1, 40, 100), torch.device('cpu'), True, True, True); print(pl.fa_bytes)"
            % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
               os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "avid-cma_amd")))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, AVID_IN_AFFINE="0"))
    assert out.returncode == 0, out.stderr[-500:]
    assert int(out.stdout.strip().splitlines()[-1]) - pl.fa_bytes >= 4 * 64 * 8 * 28 * 28 * 64 * 4
