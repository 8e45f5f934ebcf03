"""Shadows nerf/utils_wtmk_disen.py.  The reference's own file is executed as it is and every public name it defines is re-exported
(`from nerf.utils_wtmk_disen import *` in main_nerf_wtmk.py:6 is load-bearing: os, np, optim, seed_everything, the meters, Trainer); ONE
method is replaced: Trainer.train_step (utils_wtmk_disen.py:579-646) runs nerf_signature_amd.trainer.train_step -- same arguments and
return values, the ~20 stock operators around model.render / model.msg_decoder fused into this repo's kernels.  The loop around it
(train_one_epoch, GradScaler, optimiser, logging, checkpoints) stays the reference's.  NERFSIG_DROPIN_OFF=train_step: nothing is replaced."""
import importlib.util
import os
import sys

import nerf as _package
from nerf_signature_amd.switches import dropin_off as _dropin_off

_here = os.path.dirname(os.path.abspath(__file__))
_file = next((os.path.join(_p, "utils_wtmk_disen.py") for _p in _package.__path__
              if os.path.abspath(_p) != _here and os.path.isfile(os.path.join(_p, "utils_wtmk_disen.py"))), None)
if _file is None:
    raise ImportError("the reference checkout (its nerf/utils_wtmk_disen.py) has to be on sys.path behind the drop-in directory")
_spec = importlib.util.spec_from_file_location("nerf._reference_utils_wtmk_disen", _file)
_reference = importlib.util.module_from_spec(_spec)
sys.modules[_spec.name] = _reference
_spec.loader.exec_module(_reference)
globals().update({_k: _v for _k, _v in vars(_reference).items() if not _k.startswith("_")})

if not _dropin_off("train_step"):
    from nerf_signature_amd.trainer import reference_trainer_train_step as _train_step

    class Trainer(_reference.Trainer):
        train_step = _train_step
