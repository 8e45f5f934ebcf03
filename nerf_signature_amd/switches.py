"""Every environment switch the package and bench.py read -- the whole list (tests/test_host_logic.py::test_environment_switches_are_the_documented_ones
compares it with the names that occur in the sources).  Rejected experiments are not switches: their record is in LABNOTES.md / profiles/, their code is gone.

 library / arithmetic
  NERFSIG_LIB                   path of the libnerfsig.so to load (instrumented builds of the same sources: tools/build_variant.sh)
  NERFSIG_MLP                   bf16x3 | f16 (default): arithmetic of the MLP kernels at start-up (include/nerfsig.h mlp_set_precision changes it at run time)
  NERFSIG_HALF_PLANES           0: fp32 feature planes between encoder and MLP instead of the mixed fp16 layout (bit-identical; the A/B of the layout)
  NERFSIG_DECODER               torch: the HiDDeN decoder through stock torch operators instead of the fused MFMA convolutions
  NERFSIG_DETERMINISTIC         1: the codebook gradient of EVERY launch size through the fixed-point slice owners (bit-reproducible sums: csrc/hashgrid.hip k_scatter_binned);
                                by default launches below 65 536 points take the cheaper record scatter, whose float atomics make G differ by an ulp from run to run
 render paths (each pair is bit-identical; the slower form is the one the parity tests compare against)
  NERFSIG_MARCH_FUSED           0 | nf | sw: the training march as stand-alone launches (near/far, scan + write) instead of the fused ones
  NERFSIG_STAGED_FUSED          0: a staged full-image render chunk by chunk (renderer_wtmk.py:555-570 literally) instead of one fused staging pass
  NERFSIG_EVAL_LOOP             host: the eval-mode burst loop reads n_alive back every round (the reference's form) instead of device-side control
 the drop-in directory (nerf_signature_amd/dropin/: what the UNCHANGED reference CLI gets)
  NERFSIG_DROPIN_OFF            comma-separated list of accelerations to leave out: train_step (keep the reference's own Trainer.train_step), block_graph, step_graph
                                (plain eager launches instead of the captured block render / whole step), get_rays (the reference's own loader code), dense_adam
                                (the decoder's Adam step stays with torch's optimiser loop)
 more than one rank
  NERFSIG_DIST_BACKEND          gloo: rehearse N ranks on CPU tensors / on one GPU (RCCL refuses two ranks on one device)
  NERFSIG_FORCE_EXCHANGE        1: run the data-parallel exchange on a world-size-1 group (the one-GPU rehearsal of the multi-rank step)
  NERFSIG_CAPTURE_COLLECTIVES   0 | 1: RCCL collectives between captured segments | inside the step's one hipGraph (bench.py's launcher tries 1 first, then 0)
  NERFSIG_SHARD_OPTIMIZER       0 | 1: the codebook Adam replicated | sharded over the ranks (default: sharded from 4 ranks)
  NERFSIG_REPLICATE_BLOCKS      1: every rank renders all D watermark blocks (no all-gather; the launcher's last fallback)
 bench.py's launcher
  NERFSIG_LAUNCH_WATCHDOG_S     seconds an attempt of `--gpus N` may take before its ranks are killed and the next mode is tried (default 90)
  NERFSIG_SECONDARY_TIMEOUT_S   seconds one secondary child may take (default 150)
  NERFSIG_TEST_FAIL_CAPTURED    test hook: 1 | all | hang -- the first / every attempt's ranks fail or sleep on purpose (tests/test_host_logic.py)
"""
import os

DOCUMENTED = ("NERFSIG_LIB", "NERFSIG_MLP", "NERFSIG_HALF_PLANES", "NERFSIG_DECODER", "NERFSIG_DETERMINISTIC", "NERFSIG_MARCH_FUSED", "NERFSIG_STAGED_FUSED", "NERFSIG_EVAL_LOOP", "NERFSIG_DROPIN_OFF",
              "NERFSIG_DIST_BACKEND", "NERFSIG_FORCE_EXCHANGE", "NERFSIG_CAPTURE_COLLECTIVES", "NERFSIG_SHARD_OPTIMIZER", "NERFSIG_REPLICATE_BLOCKS", "NERFSIG_LAUNCH_WATCHDOG_S",
              "NERFSIG_SECONDARY_TIMEOUT_S", "NERFSIG_TEST_FAIL_CAPTURED")
DROPIN_FEATURES = ("train_step", "block_graph", "step_graph", "get_rays", "dense_adam")


def dropin_off(feature):
    """True if NERFSIG_DROPIN_OFF names `feature` (read at every call: tests flip it)."""
    if feature not in DROPIN_FEATURES:
        raise KeyError(feature)
    listed = [f.strip() for f in os.environ.get("NERFSIG_DROPIN_OFF", "").split(",") if f.strip()]
    unknown = [f for f in listed if f not in DROPIN_FEATURES]
    if unknown:
        raise ValueError(f"NERFSIG_DROPIN_OFF: unknown feature(s) {unknown}; known: {', '.join(DROPIN_FEATURES)}")
    return feature in listed


def deterministic():
    """True if NERFSIG_DETERMINISTIC=1 (read at every call)."""
    return os.environ.get("NERFSIG_DETERMINISTIC", "0") == "1"
