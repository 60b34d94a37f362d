"""CPU-side checks of the host-side mirror of the reference interface (no GPU, no kernels)."""
import numpy as np
import torch

from oracle import avid_oracle as O


def test_registry_and_state_dict_keys(golden):
    import models
    import criterions
    assert {"av_wrapper", "R2Plus1D", "Conv2D"} <= set(models.__dict__)
    assert {"AVID", "AVID_CMA"} <= set(criterions.__dict__)
    m = models.av_wrapper("R2Plus1D", {"depth": 18}, "Conv2D", {"depth": 10}, proj_dim=[512, 512, 128])
    g = golden("av_wrapper")
    sd = m.state_dict()
    assert list(sd.keys()) == list(g["state_keys"])
    assert [str(tuple(v.shape)) for v in sd.values()] == list(g["state_shapes"])
    assert m.out_dim == 128 and m.use_linear_proj
    assert sum(p.numel() for p in m.parameters()) == 21286784
    assert len(list(m.parameters())) == 141
    for d in (10, 34):
        v = models.R2Plus1D(depth=d)
        assert ["video_model." + k for k in v.state_dict()] == [n for n, _ in O.r2plus1d_spec("video_model", d)]


def test_weight_layout_survives_plumbing():
    import copy
    import models
    from avid_hip import ops
    m = models.R2Plus1D(depth=10)
    w = m.conv3x.spt_conv1.weight
    assert tuple(w.shape) == (128, 64, 1, 3, 3) and ops.weight_layout_ok(w)
    assert w.movedim(1, -1).is_contiguous()
    m2 = copy.deepcopy(m)
    m2.load_state_dict({k: v.contiguous() for k, v in m.state_dict().items()})   # plain-contiguous source tensors
    for (n, p), (_, q) in zip(m.named_parameters(), m2.named_parameters()):
        assert torch.equal(p, q) and (p.dim() < 3 or ops.weight_layout_ok(q)), n
    opt = torch.optim.Adam(m2.parameters(), lr=1e-3, weight_decay=1e-5)
    for p in m2.parameters():
        p.grad = torch.ones_like(p)
    opt.step()
    assert all(ops.weight_layout_ok(p) for p in m2.parameters() if p.dim() > 2)
    assert torch.empty_like(w).stride() == w.stride()


def test_alias_tables_match_reference(golden):
    from utils.alias_method import AliasMethod
    g = golden("alias")
    for name in ["ones999", "p4", "det50"]:
        am = AliasMethod(torch.from_numpy(g[f"{name}_probs"].copy()))
        np.testing.assert_array_equal(am.prob.numpy(), g[f"{name}_prob"])
        np.testing.assert_array_equal(am.alias.numpy(), g[f"{name}_alias"])
    assert AliasMethod(torch.ones(999)).uniform and not AliasMethod(torch.tensor([.5, .3, .1, .1])).uniform
    big = AliasMethod(torch.ones(2_000_000 - 1))      # closed form: instant (the reference loops ~1 min)
    assert big.uniform and big.alias.numel() == 1_999_999


def test_utils_namespace_extends(tmp_path, monkeypatch):
    """`utils` must merge with another `utils` directory later on sys.path (the reference's)."""
    import importlib
    import sys
    other = tmp_path / "ref" / "utils"
    other.mkdir(parents=True)
    (other / "__init__.py").write_text("")
    (other / "main_utils_probe.py").write_text("VALUE = 42\n")
    monkeypatch.syspath_prepend(str(tmp_path / "ref"))
    pkg = sys.path.pop(0)
    sys.path.append(pkg)                      # ours first, the "reference" later
    for k in [k for k in sys.modules if k == "utils" or k.startswith("utils.")]:
        monkeypatch.delitem(sys.modules, k)
    utils = importlib.import_module("utils")
    from utils.alias_method import AliasMethod  # noqa: F401  (ours)
    probe = importlib.import_module("utils.main_utils_probe")
    assert probe.VALUE == 42 and len(utils.__path__) >= 2


def test_nce_checkpoint_shape_quirk():
    from criterions.nce import NCECriterion
    c = NCECriterion(10)
    assert c.avg_exp_score.shape == ()
    c.load_state_dict({"avg_exp_score": torch.tensor([3.5])})     # reference stores shape (1,) after step 1
    assert c.avg_exp_score.shape == () and float(c.avg_exp_score) == 3.5


def test_reference_factories_resolve_build_classes():
    """The drop-in claim end to end on the host: with avid-cma_amd ahead of the reference checkout on
    sys.path, the reference's OWN factories (utils/main_utils.py:74-94, :231-238) build this repo's
    classes from the reference's own YAML config.  Skipped where the reference is absent (GPU box)."""
    import importlib
    import os
    import sys
    import pytest
    import yaml
    ref = "/root/reference"
    if not os.path.isdir(ref):
        pytest.skip("reference checkout not present")
    sys.path.append(ref)                                   # AFTER ours (conftest put avid-cma_amd first)
    try:
        for k in [k for k in sys.modules if k == "utils" or k.startswith("utils.")]:
            del sys.modules[k]
        main_utils = importlib.import_module("utils.main_utils")          # the reference's file, executed by this package's hook
        assert main_utils.REFERENCE_FILE.startswith(ref) and main_utils.build_model.__code__.co_filename.startswith(ref)
        cfg = yaml.safe_load(open(os.path.join(ref, "configs/main/avid/kinetics/Cross-N1024.yaml")))
        model = main_utils.build_model(cfg["model"])
        import models
        assert type(model).__module__ == "models.av_wrapper" and models.__file__.startswith(os.path.dirname(ref) + "/repo") \
            or "avid-cma_amd" in models.__file__
        assert model.out_dim == 128 and len(model.state_dict()) == 267
        from utils.alias_method import AliasMethod
        assert "avid-cma_amd" in sys.modules["utils.alias_method"].__file__ and AliasMethod(torch.ones(5)).uniform
        import criterions
        assert "avid-cma_amd" in criterions.__file__ and cfg["loss"]["name"] in criterions.__dict__
    finally:
        sys.path.remove(ref)
        for k in [k for k in sys.modules if k == "utils" or k.startswith("utils.")]:
            del sys.modules[k]


def test_model_checkpoint_interchange(tmp_path):
    """a reference-format checkpoint ('module.'-prefixed keys from the DataParallel wrapper,
    plain-contiguous NCDHW fp32 tensors; models/av_wrapper.py:72-74, utils/main_utils.py:265-323) loads into the
    build's channels-last parameters through ``av_wrapper(checkpoint=...)`` and saves back to tensors of the
    reference's logical shapes and values."""
    import models
    from avid_hip import ops
    src = models.av_wrapper("R2Plus1D", {"depth": 18}, "Conv2D", {"depth": 10}, proj_dim=[512, 512, 128])
    gen = torch.Generator().manual_seed(5)
    ref_sd = {}
    for k, v in src.state_dict().items():      # what the reference would have saved: logical shape, plain contiguous
        t = torch.randn(v.shape, generator=gen) if v.dtype.is_floating_point else torch.tensor(7, dtype=v.dtype)
        ref_sd["module." + k] = t.contiguous()
    path = str(tmp_path / "checkpoint.pth.tar")
    torch.save({"epoch": 3, "model": ref_sd}, path)
    m = models.av_wrapper("R2Plus1D", {"depth": 18}, "Conv2D", {"depth": 10}, proj_dim=[512, 512, 128],
                          checkpoint=path)
    back = m.state_dict()
    assert ["module." + k for k in back] == list(ref_sd)
    for k, v in back.items():
        assert v.shape == ref_sd["module." + k].shape and torch.equal(v, ref_sd["module." + k]), k
    for n, p in m.named_parameters():
        if p.dim() == 5:
            assert ops.weight_layout_ok(p), n                      # internal layout [Cout][kt][kh][kw][Cin] kept
    # and the other way round: what this build saves loads into plain (reference-layout) tensors bit-exactly
    torch.save({"model": {"module." + k: v for k, v in back.items()}}, path)
    again = torch.load(path, map_location="cpu")["model"]
    assert all(torch.equal(again["module." + k].contiguous(), ref_sd["module." + k]) for k in back)


def test_bench_launcher_argument_handling():
    """`python bench.py --gpus N` starts its own ranks (as main-avid.py:69-78 does with mp.spawn) unless it already is one."""
    import importlib.util
    import os
    import pytest
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    # a plain one-GPU run and a rank started by torch.distributed.run are not re-launched
    assert bench.launcher_command(1, ["--steps", "3"], 1, {}) is None
    assert bench.launcher_command(8, ["--gpus", "8"], 8, {"WORLD_SIZE": "8", "RANK": "3"}) is None
    cmd = bench.launcher_command(4, ["--gpus", "4", "--steps", "7"], 8, {})
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    assert cmd[-5].endswith("bench.py") and cmd[-4:] == ["--gpus", "4", "--steps", "7"]
    assert bench.launcher_command(2, [], 2, {"MASTER_PORT": "29999"})[8] == "29999"
    # the one-rank RCCL group of the 1-GPU box check goes the same way
    forced = bench.launcher_command(1, [], 1, {"AVID_FORCE_DIST": "1"})
    assert "--nproc-per-node=1" in forced
    # more ranks than GPUs: the message names the device count
    with pytest.raises(SystemExit) as e:
        bench.launcher_command(2, ["--gpus", "2"], 1, {})
    assert "shows 1 GPU" in str(e.value) and "--gpus 2" in str(e.value)
    with pytest.raises(SystemExit):
        bench.launcher_command(8, [], 0, {})


def test_utils_main_utils_is_the_references_module_with_the_two_factories_swapped(tmp_path):
    """avid-cma_amd/utils/main_utils.py executes the reference's own utils/main_utils.py (found behind it on the path) in its
    namespace and binds `torch` there to a proxy whose nn.parallel.DistributedDataParallel / optim.Adam are this build's: what
    main-avid.py:95-108 builds through utils/main_utils.py:112,250 are then the flat-buffer objects, with nothing of the reference
    edited.  Checked here on the real file when the checkout is present (build container), on a stand-in otherwise."""
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ref = "/root/reference"
    if not os.path.isfile(os.path.join(ref, "utils", "main_utils.py")):
        ref = str(tmp_path)
        os.makedirs(os.path.join(ref, "utils"))
        open(os.path.join(ref, "utils", "__init__.py"), "w").close()
        with open(os.path.join(ref, "utils", "main_utils.py"), "w") as f:     # a stand-in with the two call sites' shape
            f.write("import torch\n\ndef build_model(cfg, logger=None):\n    return None\n\n"
                    "def distribute_model_to_cuda(models, args, batch_size, num_workers, ngpus_per_node):\n"
                    "    return torch.nn.parallel.DistributedDataParallel(models, device_ids=[args.gpu]), args, batch_size, num_workers\n\n"
                    "def build_optimizer(params, cfg, logger=None):\n"
                    "    o = torch.optim.Adam(params=params, lr=cfg['lr']['base_lr'], weight_decay=cfg['weight_decay'], betas=cfg['betas'])\n"
                    "    return o, torch.optim.lr_scheduler.MultiStepLR(o, milestones=cfg['lr']['milestones'], gamma=cfg['lr']['gamma'])\n")
    code = r'''
import sys, torch
import utils.main_utils as mu
from avid_hip import parallel
assert mu.REFERENCE_FILE.startswith(sys.argv[1]), mu.REFERENCE_FILE
for name in ("build_model", "distribute_model_to_cuda", "build_optimizer"):
    f = getattr(mu, name)
    assert f.__code__.co_filename == mu.REFERENCE_FILE and f.__globals__ is vars(mu), name      # the reference's code, this namespace
on = sys.argv[2] == "1"
assert (mu.torch.optim.Adam is not torch.optim.Adam) == on and (mu.torch.nn.parallel.DistributedDataParallel is not torch.nn.parallel.DistributedDataParallel and issubclass(mu.torch.nn.parallel.DistributedDataParallel, parallel.DistributedDataParallel)) == on
assert mu.torch.optim.SGD is torch.optim.SGD and mu.torch.nn.DataParallel is torch.nn.DataParallel and mu.torch.save is torch.save
cfg = {"name": "adam", "lr": {"base_lr": 2e-4, "milestones": [10], "gamma": 1.0}, "weight_decay": 1e-5, "betas": [0.9, 0.999]}
# a bare model's parameters (the evaluation scripts' case, eval-action-recg.py:58,76): torch's own Adam either way
lin = torch.nn.Linear(8, 4)
opt, sched = mu.build_optimizer(lin.parameters(), cfg)
assert type(opt) is torch.optim.Adam and isinstance(sched, torch.optim.lr_scheduler.MultiStepLR) and opt.param_groups[0]["lr"] == 2e-4
assert type(mu.build_optimizer([lin.weight], cfg)[0]) is torch.optim.Adam
# the parameters of a model whose flat buffers a wrapper / engine holds (main-avid.py:93-108): the flat-buffer Adam
net = torch.nn.Sequential(torch.nn.Linear(8, 8), torch.nn.Linear(8, 4))
flat = parallel.FlatParams(net)
opt, _ = mu.build_optimizer(list(net.parameters()), cfg)
assert (type(opt) is parallel.Adam and opt.flat is flat) if on else type(opt) is torch.optim.Adam
assert type(mu.build_optimizer([net[0].weight], cfg)[0]) is torch.optim.Adam          # a subset of it: torch's, nothing re-seated
assert all(p.data_ptr() == flat.flat.data_ptr() + 4 * o for p, o in zip(flat.params, flat.offsets))
print("OK")
'''
    for on in ("1", "0"):
        env = dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.join(repo, "avid-cma_amd"), ref]), AVID_DROPIN=on)
        out = subprocess.run([sys.executable, "-c", code, ref, on], capture_output=True, text=True, env=env, cwd=str(tmp_path))
        assert out.returncode == 0 and "OK" in out.stdout, out.stderr[-1500:]


def test_a_cu_budget_below_the_column_block_count_still_plans(tmp_path):
    """with avid_set_cu_budget(8) and a layer of 16 column blocks, plan_pk_tile's `cus / ntn * ntn` was 0 and the next
    line divided by it (SIGFPE).  The planner now deals at least one M-tile's worth of column blocks.  Host code only: no GPU."""
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import sys, ctypes as C
sys.path.insert(0, sys.argv[1])
from avid_hip import lib, ops
assert lib.raw("avid_set_cu_budget")(8) == 8
buf = C.create_string_buffer(256)
for cin, cout in ((64, 1024), (128, 1024), (512, 512)):
    d = ops._desc((4, 1, 32, 32), cin, cout, (1, 1, 1), (1, 1, 1), (0, 0, 0), False)
    for which in (0, 1):
        assert lib.raw("avid_conv_kernel_name")(C.byref(d), which, buf, 256) == 0 and buf.value.startswith(b"igemm_pk_kernel")
    assert lib.raw("avid_conv_fwd_workspace_bytes")(C.byref(d)) >= 0 and lib.raw("avid_conv_fwd_stats_rows")(C.byref(d)) >= 0
print("OK")
'''
    out = subprocess.run([sys.executable, "-c", code, os.path.join(repo, "avid-cma_amd")], capture_output=True, text=True, cwd=str(tmp_path))
    assert out.returncode == 0 and "OK" in out.stdout, (out.returncode, out.stderr[-800:])


def test_dropin_adam_refuses_a_parameter_list_that_is_not_its_flat_buffers():
    """Adam over a list that shares parameters with a live flat buffer without being exactly its set used to re-seat them
    into a second buffer (the launch programs then wrote one buffer and the optimizer read the other: no learning, silently)."""
    import pytest
    from avid_hip import parallel
    lin = torch.nn.Linear(8, 8)
    extra = torch.nn.Parameter(torch.zeros(3))
    flat = parallel.FlatParams(lin)
    opt = parallel.Adam(lin.parameters(), lr=1e-3)
    assert opt.flat is flat                                           # exactly the set: adopted
    with pytest.raises(ValueError, match="not exactly that buffer's parameter set"):
        parallel.Adam(list(lin.parameters()) + [extra], lr=1e-3)      # superset
    with pytest.raises(ValueError, match="not exactly that buffer's parameter set"):
        parallel.Adam([lin.weight], lr=1e-3)                          # subset
    other = torch.nn.Linear(4, 4)
    assert parallel.Adam(other.parameters(), lr=1e-3).flat is not flat    # untouched parameters: flattened here


def test_grad_buckets_reset_drops_what_a_failed_backward_left_behind():
    """per-step state of the bucketed all-reduce is restored at the start of a training forward of the wrapper
    (GradBuckets.reset): counts a raised backward pass left behind must not keep the next step's buckets from launching."""
    from avid_hip import parallel
    lin = torch.nn.Sequential(torch.nn.Linear(8, 8), torch.nn.Linear(8, 4))
    flat = parallel.FlatParams(lin)
    b = parallel.GradBuckets(flat, bucket_bytes=64)
    counts = list(b.counts)
    b.pending[0] -= 1                       # as if one gradient of bucket 0 had been reported when the pass died
    b.launched[-1] = True
    b.reported[0] = True
    b.reset()
    assert b.pending == counts and not any(b.launched) and not any(b.reported) and b.works == [] and b.step_set is None
