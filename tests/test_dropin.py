"""The drop-in directory shadows exactly the hot-path modules by name; everything else still resolves to the
reference checkout.  Needs /root/reference (build container only) -- skipped elsewhere."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"

SCRIPT = r'''
import sys, types
from unittest.mock import MagicMock
for name in ("trimesh","cv2","imageio","tensorboardX","mcubes","torch_ema","lpips","torchmetrics","torchmetrics.functional",
             "matplotlib","matplotlib.pyplot","torchvision","torchvision.transforms","scipy.spatial.transform","PIL","PIL.Image"):
    try: __import__(name)
    except Exception: sys.modules[name] = MagicMock()
import nerf, nerf.network_wtmk_tcnn, nerf.renderer_wtmk, raymarching, hash_encoding, hash_encoding_wtmk_bit, activation
import nerf_signature_amd.network as ours
assert issubclass(nerf.network_wtmk_tcnn.NeRFNetwork, ours.NeRFNetwork)      # the same model, with the foreign-loop accelerations switched on
_m = nerf.network_wtmk_tcnn.NeRFNetwork(bound=1.0, cuda_ray=True, message_dim=4, n_views=1)
assert _m.shared_gradient_step and _m.auto_fix_rays and set(_m.state_dict()) == set(ours.NeRFNetwork(bound=1.0, cuda_ray=True, message_dim=4, n_views=1).state_dict())
assert hash_encoding.HashEmbedder.__module__ == "nerf_signature_amd.hash_encoding"
assert hash_encoding_wtmk_bit.HashEmbedder.__module__ == "nerf_signature_amd.hash_encoding_wtmk_bit"
assert raymarching.__file__.startswith(ROOT) and callable(raymarching.march_rays_train)
import nerf.utils_wtmk_disen as u                       # the reference's own trainer module, executed as it is, with Trainer.train_step replaced
import nerf._reference_utils_wtmk_disen as ref_u
assert ref_u.__file__.startswith(REF) and u.__file__.startswith(ROOT), (ref_u.__file__, u.__file__)
assert hasattr(u, "Trainer") and hasattr(u, "BIT_ACC") and hasattr(u, "seed_everything")
for name in ("os", "np", "optim", "seed_everything", "PSNRMeter", "LPIPSMeter", "SSIMMeter", "BIT_ACC", "get_rays"):      # what main_nerf_wtmk.py takes through `import *`
    assert getattr(u, name) is getattr(ref_u, name), name
import nerf_signature_amd.trainer as our_trainer
assert u.Trainer.__mro__[1] is ref_u.Trainer and u.Trainer.train_step is our_trainer.reference_trainer_train_step
assert {k for k in vars(u.Trainer) if not k.startswith("__")} == {"train_step"}       # nothing else of the Trainer is touched
import nerf.provider_wtmk as prov                       # another module of the reference: still its own file ...
assert prov.__file__.startswith(REF)
import nerf.utils_wtmk as uw, nerf._reference_utils_wtmk as ref_uw, nerf_signature_amd.rays as our_rays      # ... which takes get_rays from the shadowed nerf.utils_wtmk
assert uw.__file__.startswith(ROOT) and ref_uw.__file__.startswith(REF) and prov.get_rays is uw.get_rays and uw.get_rays is not ref_uw.get_rays
assert {k for k in vars(ref_uw) if not k.startswith("_") and getattr(uw, k, None) is not getattr(ref_uw, k)} == {"get_rays"}
import torch
_poses = torch.eye(4)[None]
_intr = [50.0, 50.0, 8.0, 8.0]
torch.manual_seed(3); a = uw.get_rays(_poses, _intr, 16, 16, 10)          # CPU poses: the reference's own function
torch.manual_seed(3); b = ref_uw.get_rays(_poses, _intr, 16, 16, 10)
assert sorted(a) == sorted(b) and all(torch.equal(a[k], b[k]) for k in a)
# main_nerf_wtmk.py:93-102,110: the model is built and its optimiser groups taken exactly as the CLI does
m = nerf.network_wtmk_tcnn.NeRFNetwork(bound=1.0, cuda_ray=True, density_scale=1, min_near=0.2, density_thresh=10, bg_radius=-1, message_dim=32, n_views=1)
import torch
opt = torch.optim.Adam(m.get_params(1e-2), betas=(0.9, 0.99), eps=1e-15)
assert len(opt.param_groups) == 2
# main_nerf_wtmk.py:110-119: the reference's OWN Trainer around this repo's model (no render on the CPU: construction, its parameter
# count, its save_checkpoint / load_checkpoint code on our state_dict, and a fresh model resumed from that file by the reference's code)
import argparse, os, tempfile
ws = tempfile.mkdtemp()
o = argparse.Namespace(lambda_w=1.0, lambda_i=1.0, distortion="none", loss_w="bce", patch_size=1, rand_pose=-1, error_map=False, color_space="srgb")
optimizer = lambda model: torch.optim.Adam(model.get_params(1e-2), betas=(0.9, 0.99), eps=1e-15)
tr = u.Trainer("ngp", o, m, optimizer=optimizer, device=torch.device("cpu"), workspace=ws, fp16=False, use_checkpoint="scratch", use_tensorboardX=False,
               message_dim=32, n_views=1, metrics=[u.PSNRMeter()], metrics_message=[u.BIT_ACC(device="cpu")], mute=True)
assert tr.model is m and isinstance(tr.optimizer, torch.optim.Adam)
with torch.no_grad():
    m.msg_encoder.embeddings[5].weight.uniform_(-1, 1)
    m.density_bitfield.fill_(9)
tr.epoch, tr.global_step = 3, 77
tr.save_checkpoint(full=True, best=False)
files = sorted(os.listdir(os.path.join(ws, "checkpoints")))
assert files == ["ngp_ep0003.pth"], files
m2 = nerf.network_wtmk_tcnn.NeRFNetwork(bound=1.0, cuda_ray=True, density_scale=1, min_near=0.2, density_thresh=10, bg_radius=-1, message_dim=32, n_views=1)
tr2 = u.Trainer("ngp", o, m2, optimizer=optimizer, device=torch.device("cpu"), workspace=ws, fp16=False, use_checkpoint="latest", use_tensorboardX=False,
                message_dim=32, n_views=1, mute=True)
assert tr2.epoch == 3 and tr2.global_step == 77
for (k1, v1), (k2, v2) in zip(m.state_dict().items(), m2.state_dict().items()):
    assert k1 == k2 and torch.equal(v1, v2), k1
print("DROPIN_OK")
'''


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout not present")
def test_dropin_shadows_only_the_hot_path_modules(tmp_path):
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1",
               PYTHONPATH=os.pathsep.join([os.path.join(ROOT, "nerf_signature_amd", "dropin"), ROOT, REF]))
    script = SCRIPT.replace("REF", repr(REF)).replace("ROOT", repr(ROOT))
    out = subprocess.run([sys.executable, "-B", "-c", script], env=env, capture_output=True, text=True, cwd=str(tmp_path), timeout=300)
    assert "DROPIN_OK" in out.stdout, out.stderr[-3000:]


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout not present")
def test_dropin_trainer_shadow_can_be_switched_off(tmp_path):
    """NERFSIG_DROPIN_OFF=train_step: the shadow module re-exports the reference's Trainer itself."""
    script = r'''
import sys
from unittest.mock import MagicMock
for name in ("trimesh","cv2","imageio","tensorboardX","mcubes","torch_ema","lpips","torchmetrics","torchmetrics.functional",
             "matplotlib","matplotlib.pyplot","torchvision","torchvision.transforms","scipy.spatial.transform","PIL","PIL.Image"):
    try: __import__(name)
    except Exception: sys.modules[name] = MagicMock()
import nerf.utils_wtmk_disen as u
import nerf._reference_utils_wtmk_disen as ref_u
assert u.Trainer is ref_u.Trainer
print("DROPIN_OK")
'''
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1", NERFSIG_DROPIN_OFF="train_step",
               PYTHONPATH=os.pathsep.join([os.path.join(ROOT, "nerf_signature_amd", "dropin"), ROOT, REF]))
    out = subprocess.run([sys.executable, "-B", "-c", script], env=env, capture_output=True, text=True, cwd=str(tmp_path), timeout=300)
    assert "DROPIN_OK" in out.stdout, out.stderr[-3000:]
