"""Shadows the reference's nerf/hidden_models.py (decoder on stock PyTorch ops; no torchvision needed)."""
from nerf_signature_amd.hidden_models import (ConvBNRelu, HiddenDecoder_multi_views, get_hidden_decoder_multi_views,  # noqa: F401
                                              normalize_img, unnormalize_img)
