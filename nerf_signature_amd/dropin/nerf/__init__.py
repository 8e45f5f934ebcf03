"""Shadows the reference's `nerf` package for the modules on the hot path only.

The modules in this directory replace nerf/network_wtmk_tcnn.py, nerf/renderer_wtmk.py and
nerf/hidden_models.py; every other `nerf.<module>` (utils_wtmk_disen, provider_wtmk, ...) is looked up in the
reference checkout's own nerf/ directory, found further down sys.path, so those files are used as they are."""
import os
import sys

for _p in sys.path:
    _cand = os.path.join(_p, "nerf")
    if os.path.isdir(_cand) and os.path.abspath(_cand) != os.path.dirname(os.path.abspath(__file__)) and _cand not in __path__:
        __path__.append(_cand)
