import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The oracle's tensor ops are small; on a many-core host (a GPU box reports 256 logical CPUs of which a job may use a share) torch's default
    # of one intra-op thread per logical CPU oversubscribes them by an order of magnitude (the trajectory test: 267 s against ~35 s on 8 cores).
    import torch
    allowed = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:      # the container's CPU share (cgroup v2)
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        allowed = allowed if q == "max" else min(allowed, max(1, int(int(q) / int(per))))
    except Exception:
        pass
    torch.set_num_threads(max(1, min(allowed, 16)))


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(params=["bf16x3", "f16"])
def mlp_prec(request):
    """Run a GPU test under both arithmetics of the MLP kernels (include/nerfsig.h: mlp_set_precision) and restore the default."""
    from nerf_signature_amd import _native as nv
    before = nv.fn("mlp_get_precision")()
    nv.set_mlp_precision(request.param)
    yield request.param
    nv.call("mlp_set_precision", before)


@pytest.fixture
def strict_mlp():
    """fp32-class arithmetic (split-bf16) for tests that compare GRADIENTS element-wise with the fp32 oracle: at fp16 operand precision
    ~6 % of the points have some ReLU pre-activation so close to zero that the kernel and the oracle take different sides of the kink,
    and such a point's gradient differs by a finite amount whatever the accumulate precision (tests/test_gpu_field.py::
    test_field_backward_away_from_relu_kinks quantifies this; the reference's tinycudann MLP is fp16 with fp16 accumulate)."""
    from nerf_signature_amd import _native as nv
    before = nv.fn("mlp_get_precision")()
    nv.set_mlp_precision("bf16x3")
    yield
    nv.call("mlp_set_precision", before)
