"""The training step of the watermark stage: Trainer.train_step and the loop body around it
(/root/reference/nerf/utils_wtmk_disen.py:579-646, 1164-1190), plus the BIT_ACC / PSNR meters (:211-245, :321-361).

`train_step` reproduces the reference's arithmetic for its default configuration (3-channel images, srgb,
distortion 'none', loss_w 'bce'): a block render decoded into message logits, a content render compared
with the clean image, loss = lambda_w * BCE(10 * logits, message) + lambda_i * MSE.
`WatermarkLoop.step` is the loop body: zero grads, train_step, backward, gradient exchange, optimiser step --
with no host synchronisation of its own (losses stay on the device)."""
import numpy as np
import torch
import torch.nn.functional as F

from . import fieldops as fo
from .dp import GradExchange


def loss_w_bce(decoded, keys, temp=10.0):
    return F.binary_cross_entropy_with_logits(decoded * temp, keys, reduction="mean")


def loss_w_mse(decoded, keys, temp=10.0):
    return torch.mean((decoded * temp - (2 * keys - 1)) ** 2)


def train_step(model, data, message, render_kwargs, lambda_w=1.0, lambda_i=1.0, loss_w=loss_w_bce):
    """data = {'watermark': {'rays_o_block', 'rays_d_block'}, 'content': {'rays_o', 'rays_d', 'images'}}.
    Returns (pred_rgb, gt_rgb, content_pred_rgb, lossi, lossw, loss) like the reference."""
    wm, content = data["watermark"], data["content"]
    kw = dict(render_kwargs)
    kw.update(staged=False, bg_color=1, perturb=False, force_all_rays=True)
    outputs = model.render(wm["rays_o_block"], wm["rays_d_block"], message, **kw)
    pred_rgb = torch.clamp(outputs["image"], min=0, max=1)
    decoded = model.msg_decoder(model.normalization(pred_rgb.permute(0, 3, 1, 2)))
    gt_rgb = content["images"]
    content_pred_rgb = model.render(content["rays_o"], content["rays_d"], message, **kw)["image"]
    lossi = ((content_pred_rgb - gt_rgb) ** 2).mean()
    lossw = loss_w(decoded, message.to(decoded.device).unsqueeze(-1))
    loss = lambda_w * lossw + lambda_i * lossi
    return pred_rgb, gt_rgb, content_pred_rgb, lossi, lossw, loss


class WatermarkLoop:
    """Loop body of train_one_epoch (utils_wtmk_disen.py:1164-1181) for one model replica."""

    def __init__(self, model, optimizer, render_kwargs, lambda_w=1.0, lambda_i=1.0, lr_scheduler=None, use_sink=True):
        self.model, self.optimizer, self.lr_scheduler = model, optimizer, lr_scheduler
        self.render_kwargs = dict(render_kwargs)
        self.lambda_w, self.lambda_i = lambda_w, lambda_i
        dev = next(model.parameters()).device
        self.sink = fo.GradSink(dev) if use_sink else None
        model.grad_sink = self.sink
        self.exchange = GradExchange(list(model.msg_decoder.parameters()))
        self.last = None

    def step(self, data, message):
        """message: float tensor of 0./1. (keep it on the CPU to avoid the D2H read of its bits)."""
        self.optimizer.zero_grad(set_to_none=True)
        if self.sink is not None:
            self.sink.zero_()
        out = train_step(self.model, data, message, self.render_kwargs, self.lambda_w, self.lambda_i)
        out[-1].backward()
        self.exchange(self.sink.G if self.sink is not None else None)
        if self.sink is not None and hasattr(self.optimizer, "step_shared"):
            self.optimizer.step_shared(self.sink.selected, self.sink.G)   # fused: no per-table gradients are materialised
        elif self.sink is not None:
            self.sink.fanout()
        self.optimizer.step()
        if self.lr_scheduler is not None:
            self.lr_scheduler.step()
        self.last = out
        return out


class BIT_ACC:
    """Fraction of message bits whose decoded sign is right (utils_wtmk_disen.py:321-361)."""

    def __init__(self, device=None):
        self.V, self.N, self.instant_V = 0, 0, 0

    def clear(self):
        self.V, self.N = 0, 0

    def update(self, preds, truths):
        diff = ~torch.logical_xor(preds > 0, truths > 0)
        acc = (torch.sum(diff, dim=-1) / diff.shape[-1]).item()
        self.instant_V = acc
        self.V += acc
        self.N += 1

    def measure(self):
        return self.V / self.N

    def report(self):
        return f"bit_acc = {self.measure():.6f}"


class PSNRMeter:
    """-10 log10(MSE) averaged over updates (utils_wtmk_disen.py:211-245)."""

    def __init__(self):
        self.V, self.N = 0, 0

    def clear(self):
        self.V, self.N = 0, 0

    def update(self, preds, truths):
        p = preds.detach().cpu().numpy() if torch.is_tensor(preds) else preds
        t = truths.detach().cpu().numpy() if torch.is_tensor(truths) else truths
        self.V += -10 * np.log10(np.mean((p - t) ** 2))
        self.N += 1

    def measure(self):
        return self.V / self.N

    def report(self):
        return f"PSNR = {self.measure():.6f}"
