import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, "avid-cma_amd")
for p in (REPO, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return load


@pytest.fixture(scope="session")
def gpu_device():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


@pytest.fixture(params=["default", "wino_forced", "wino2_forced"])
def wino_mode(request, gpu_device):
    """"wino_forced": every (1,3,3) stride-1 layer with Cin % 32 == 0 and 64 | Cout <= 128 goes through the
    Winograd kernels whatever its size (avid_wino_configure), so that the small reference-generated fixtures
    traverse them; "wino2_forced": the same through ``wino2_kernel`` (avid_wino2_configure), which otherwise only
    takes layers with many tiles.  The tests assert from the launch log that they did."""
    from avid_hip import ops
    if request.param != "default":
        ops.wino_configure(1, 1, 128)
    if request.param == "wino2_forced":
        ops.wino2_configure(0)
    yield request.param
    ops.wino_configure(-1, -1, -1)
    ops.wino2_configure(-1)


@pytest.fixture(params=["auto", "wino2"])
def wino_variant(request, gpu_device):
    """"wino2": layers on the Winograd path run on ``wino2_kernel`` whatever their size (the dispatch rule stays)."""
    from avid_hip import ops
    if request.param == "wino2":
        ops.wino2_configure(0)
    yield request.param
    ops.wino2_configure(-1)


class _KernelLog:
    """HIP-event timers of the library used as a launch log: which kernels ran inside the ``with`` block."""

    def __enter__(self):
        from avid_hip import lib
        self.lib = lib
        lib.timing_enable(True)
        self.report = {}
        return self

    def __exit__(self, *exc):
        import torch
        torch.cuda.synchronize()
        self.report = self.lib.timing_report()
        self.lib.timing_enable(False)
        return False

    def launches(self, prefix):
        return sum(v["launches"] for k, v in self.report.items() if k.startswith(prefix))


@pytest.fixture
def kernel_log():
    return _KernelLog


@pytest.fixture
def per_layer_path(monkeypatch):
    """The test exercises the per-layer autograd path of the step engine (avid_hip/ops.py: what a hooked / partly frozen
    model falls back to), not the compiled launch programs (avid_hip/plan.py), which are the default."""
    from avid_hip import plan
    monkeypatch.setattr(plan, "ENABLED", False)
