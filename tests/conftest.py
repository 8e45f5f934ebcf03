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


# ---- order of the GPU suite -------------------------------------------------------------------------------------------------------------------------
# The driver runs `pytest tests -x -q -m gpu`: the first failure ends the run, so everything behind it is unrecorded.  The suite therefore runs from
# the sharpest evidence to the softest: (0) kernels against the reference's own native module, the golden fixtures and the CPU oracle (bit-exact or
# element-wise), (1) glue, losses, optimiser passes and the stage-1 gradients against the oracle, (2) the full-size workloads, and LAST (3) every test
# that trains for hundreds of steps and asserts on a learned quantity (PSNR, bit accuracy).  Inside a tier the files keep the order below and the tests
# their order in the file.
_TIER0_FILES = ["test_gpu_ref_native.py", "test_gpu_raymarch.py", "test_gpu_field.py", "test_gpu_grid.py", "test_gpu_edges.py",
                "test_gpu_tcnn_gap.py", "test_gpu_distortion.py"]
_TIER1_FILES = ["test_gpu_render.py", "test_gpu_stage1.py", "test_gpu_fixed.py", "test_gpu_amp_ckpt.py", "test_gpu_dp.py"]
_TIER2_FILES = ["test_gpu_fullsize.py"]
_TIER3_FILES = ["test_gpu_convergence.py"]
# kernel-vs-stock-operator tests that live in test_gpu_render.py: tier 0 (R13 decoder, Adam passes)
_TIER0_NAMES = ("test_fused_batchnorm_gelu", "test_fused_decoder_matches", "test_fused_finish_and_loss", "test_fused_decoder_on_rendered",
                "test_dense_adam", "test_codebook_adam", "test_fused_decoder_gradients_are_views")
# multi-hundred-step training / statistical tests that live outside test_gpu_convergence.py: tier 3
_TIER3_NAMES = ("test_training_trajectory_psnr_and_bit_accuracy", "test_uniform_sample_path_trains_the_codebook", "test_finetune_decoder_mode",
                "test_other_message_lengths_train_step_and_captured_loop", "test_captured_loop_tracks_the_cpu_oracle_over_200_steps",
                "test_fixed_block_cache_trains_like_the_loop_that_recomputes", "test_grid_refresh_runs_in_its_cadence", "test_grid_refresh_full_and_partial",
                "test_geometric_distortions_run_in_the_eager_loop")


# inside tier 3: runs a CPU oracle tracks step for step first, cold-start runs judged by a learned statistic at the very end
_TIER3_LAST = ("test_bench_size_training_converges", "test_the_other_configs_learn", "test_robustness_training", "test_the_bound_trainer_train_step_trains")


def gpu_suite_rank(path, name):
    """(tier, position inside the tier) of one GPU test; `path` is the file's base name, `name` the test function's name."""
    files = _TIER0_FILES + _TIER1_FILES + _TIER2_FILES + _TIER3_FILES
    pos = files.index(path) if path in files else len(files)
    if name.startswith(_TIER3_NAMES) or path in _TIER3_FILES:
        return 3, pos + (100 if name.startswith(_TIER3_LAST) else 0)
    if name.startswith(_TIER0_NAMES) or path in _TIER0_FILES:
        return 0, pos
    if path in _TIER2_FILES:
        return 2, pos
    return 1, pos


def pytest_collection_modifyitems(config, items):
    keyed = []
    for i, it in enumerate(items):
        if it.get_closest_marker("gpu") is None:
            keyed.append(((-1, 0), i, it))                 # CPU tests: untouched, in front
        else:
            keyed.append((gpu_suite_rank(os.path.basename(str(it.fspath)), it.originalname or it.name), i, it))
    keyed.sort(key=lambda t: (t[0], t[1]))
    items[:] = [t[2] for t in keyed]
