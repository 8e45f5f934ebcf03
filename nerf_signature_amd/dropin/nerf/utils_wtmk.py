"""Shadows nerf/utils_wtmk.py -- the module the reference's dataset takes `get_rays` from (provider_wtmk.py:16).  The reference's own file is executed
as it is and every public name re-exported; ONE function is replaced: get_rays (utils_wtmk.py:57-143) for poses that live on the GPU is
nerf_signature_amd.rays.get_rays -- the same arguments, the same random draws (torch.randint / multinomial / rand in the same order, so the same pixel
indices), the same dict of results, the rays themselves from one launch (rg_get_rays: origins bit-identical, directions within 5e-7 of the reference's
~25 small torch launches; tests/test_gpu_raymarch.py::test_get_rays_on_device_matches_reference_golden).  CPU poses (no --preload) go to the
reference's function.  NERFSIG_DROPIN_OFF=get_rays: nothing is replaced."""
import importlib.util
import os
import sys

import nerf as _package
from nerf_signature_amd.switches import dropin_off as _dropin_off

_here = os.path.dirname(os.path.abspath(__file__))
_file = next((os.path.join(_p, "utils_wtmk.py") for _p in _package.__path__
              if os.path.abspath(_p) != _here and os.path.isfile(os.path.join(_p, "utils_wtmk.py"))), None)
if _file is None:
    raise ImportError("the reference checkout (its nerf/utils_wtmk.py) has to be on sys.path behind the drop-in directory")
_spec = importlib.util.spec_from_file_location("nerf._reference_utils_wtmk", _file)
_reference = importlib.util.module_from_spec(_spec)
sys.modules[_spec.name] = _reference
_spec.loader.exec_module(_reference)
globals().update({_k: _v for _k, _v in vars(_reference).items() if not _k.startswith("_")})

if not _dropin_off("get_rays"):
    from nerf_signature_amd.rays import get_rays as _device_get_rays

    def get_rays(poses, intrinsics, H, W, N=-1, error_map=None, patch_size=1):
        if getattr(poses, "is_cuda", False):
            return _device_get_rays(poses, intrinsics, H, W, N, error_map, patch_size)
        return _reference.get_rays(poses, intrinsics, H, W, N, error_map, patch_size)
