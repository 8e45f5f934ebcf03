"""The training step of the watermark stage: Trainer.train_step and the loop body around it
(/root/reference/nerf/utils_wtmk_disen.py:579-646, 1164-1190), plus the BIT_ACC / PSNR meters (:211-245, :321-361).

`train_step` reproduces the reference's arithmetic for its default configuration (3-channel images, srgb,
distortion 'none', loss_w 'bce'): a block render decoded into message logits, a content render compared
with the clean image, loss = lambda_w * BCE(10 * logits, message) + lambda_i * MSE.
`WatermarkLoop.step` is the loop body: zero grads, train_step, backward, gradient exchange, optimiser step --
with no host synchronisation of its own (losses stay on the device)."""
import ctypes
import os

import numpy as np
import torch
import torch.nn.functional as F

from . import _native as nv
from . import fieldops as fo
from . import dp
from .dp import GradExchange, exchange_active, world_size
from .distortion import DistortionLayer, host_uniform, scaled_width
from .hidden_models import normalize_img, set_grad_arena, set_weights_stream


def loss_w_bce(decoded, keys, temp=10.0):
    return F.binary_cross_entropy_with_logits(decoded * temp, keys, reduction="mean")


def loss_w_mse(decoded, keys, temp=10.0):
    return torch.mean((decoded * temp - (2 * keys - 1)) ** 2)


class _WatermarkLoss(torch.autograd.Function):
    """(lossi, lossw, loss) of train_step -- MSE, BCE-with-logits at temperature 10, weighted sum -- as one kernel each way
    (wm_loss_fwd / wm_loss_bwd) instead of ~25 stock elementwise/reduction launches."""

    @staticmethod
    def forward(ctx, content, gt, decoded, keys, lambda_w, lambda_i, temp):
        content, gt, decoded, keys = content.contiguous(), gt.contiguous(), decoded.contiguous(), keys.contiguous()
        losses = torch.empty(3, dtype=torch.float32, device=content.device)
        d_content, d_decoded = torch.empty_like(content), torch.empty_like(decoded)
        nv.call("wm_loss_fwd", nv.ptr(content), nv.ptr(gt), content.numel(), nv.ptr(decoded), nv.ptr(keys), decoded.numel(), temp, lambda_w, lambda_i,
                nv.ptr(losses), nv.ptr(d_content), nv.ptr(d_decoded), nv.stream())
        ctx.save_for_backward(d_content, d_decoded)
        ctx.lambdas = (lambda_w, lambda_i)
        ctx.set_materialize_grads(False)
        _WatermarkLoss.stash = (d_content, d_decoded)   # see backward_from_loss_kernel
        return losses[0], losses[1], losses[2]

    @staticmethod
    def backward(ctx, g_li, g_lw, g_l):
        d_content, d_decoded = ctx.saved_tensors
        g_content, g_decoded = torch.empty_like(d_content), torch.empty_like(d_decoded)
        nv.call("wm_loss_bwd", *(nv.ptr(None if g is None else g.contiguous()) for g in (g_li, g_lw, g_l)), *ctx.lambdas, nv.ptr(d_content),
                d_content.numel(), nv.ptr(d_decoded), d_decoded.numel(), nv.ptr(g_content), nv.ptr(g_decoded), nv.stream())
        return g_content, None, g_decoded, None, None, None, None


def backward_from_loss_kernel(out, content_scale=1.0, content_stream=None, content_first=False):
    """`out[-1].backward()` for a train_step whose losses came from wm_loss_fwd, without the ones-fill and the wm_loss_bwd launch:
    the forward kernel already left d(loss_i)/d(content) and d(loss_w)/d(decoded); for an upstream gradient of 1 they only need
    the lambdas, which are applied here as the (host-side) scale of the seed.
    content_scale: extra factor on the content loss's seed (dp.content_grad_scale: 1/world when the watermark blocks are sharded
    over the ranks, so that the gradient exchange is a plain sum)."""
    last = getattr(_WatermarkLoss, "last", None)
    _WatermarkLoss.last = None
    if last is None or last[0] is not out[-1]:
        if content_scale == 1.0:
            out[-1].backward()
        else:                       # loss = lambda_w * lossw + lambda_i * lossi (train_step): re-weight the image term
            lambda_w, lambda_i = getattr(train_step, "last_lambdas", (1.0, 1.0))
            (lambda_w * out[4] + (lambda_i * content_scale) * out[3]).backward()
        return
    _, content, decoded, d_content, d_decoded, lambda_w, lambda_i, fused_seed = last
    ci = lambda_i * content_scale
    # the decoder's seed: lambda_w * d(loss_w)/d(decoded) -- left behind by the decoder's own head kernel where it could (train_step), else from the loss kernel
    seeds = [None if not content.requires_grad else (d_content if ci == 1.0 else d_content * ci),
             fused_seed if fused_seed is not None else (d_decoded if lambda_w == 1.0 else d_decoded * lambda_w)]
    # (finetune_decoder: the codebook is frozen too, the content render then has no trainable input and no grad_fn)
    pairs = [(t, g) for t, g in zip((content, decoded), seeds) if t.requires_grad]
    if len(pairs) == 2 and content_stream is not None and content_first:
        # Two disjoint autograd graphs (they meet only in the shared gradient G, a side effect).  In one backward call over both, the content
        # render's backward -- whose nodes are the oldest, so the engine runs them last -- starts only when everything the main stream holds by
        # then (the decoder's and the block render's backward) is done, and queues behind the decoder's parameter gradients on its own stream:
        # it ends up at the step's tail (kernel timelines: profiles/r03_backward_schedule.txt).  Issued first, from its own stream (the engine
        # joins the streams a call used into the CALLER's stream when it returns), it starts right behind the loss kernel and runs beside the
        # decoder's backward chain.
        content_stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(content_stream):
            torch.autograd.backward([pairs[0][0]], [pairs[0][1]])
        torch.autograd.backward([pairs[1][0]], [pairs[1][1]])
    elif pairs:
        torch.autograd.backward([t for t, _ in pairs], [g for _, g in pairs])


def join_stream(main, other):
    """main.wait_stream(other) -- except while `main` is capturing and `other` was never forked into that capture (it then holds no work of the
    step, and a captured stream must not wait for an event recorded outside its capture)."""
    if other is None or other is main:
        return
    if torch.cuda.is_current_stream_capturing():
        with torch.cuda.stream(other):
            forked = torch.cuda.is_current_stream_capturing()
        if not forked:
            return
    main.wait_stream(other)


def local_blocks(wm):
    """The watermark-block rays this rank renders: all of them, or its shard (dp.block_shard) as views of the same tensors.
    Returns (rays_o, rays_d, shard or None)."""
    o, d = wm["rays_o_block"], wm["rays_d_block"]
    shard = dp.block_shard(o.shape[0]) if o.dim() == 4 else None
    if shard is None:
        return o, d, None
    return o[shard[0]:shard[1]], d[shard[0]:shard[1]], shard


def srgb_to_linear(x):
    """utils_wtmk_disen.py:54-55."""
    return torch.where(x < 0.04045, x / 12.92, ((x + 0.055) / 1.055) ** 2.4)


def train_step(model, data, message, render_kwargs, lambda_w=1.0, lambda_i=1.0, loss_w=loss_w_bce, side_stream=None, presum_first=False,
               presum_adopt=False, color_space="srgb", blocks_first=False, content_backward_now=None, distortion=None, block_graph=None):
    """data = {'watermark': {'rays_o_block', 'rays_d_block'}, 'content': {'rays_o', 'rays_d', 'images'}}.
    Returns (pred_rgb, gt_rgb, content_pred_rgb, lossi, lossw, loss) like the reference.

    distortion: the reference's `--distortion` (utils_wtmk_disen.py:551-577,594): a name (none | noise | rotation | scaling | blurring |
    brightness; this call then draws the step's random parameters) or a distortion.DistortionLayer whose owner has drawn them (the loops).
    block_graph: a blockgraph.BlockDecodeGraph (eager callers with constant block rays: the drop-in Trainer.train_step) -- the block render and the
    decoder, forward and backward, replayed as two captured graphs once the rays are kept; it declines (and the eager launches run) whenever it does not apply.

    side_stream: a torch.cuda.Stream on which the content render is issued.  It depends on nothing the block render or the
    decoder produce, and both chains are sequences of small latency-bound launches, so they overlap (forward and -- autograd
    replays a node on the stream it ran on -- backward); the shared codebook gradient is accumulated with atomics by both."""
    wm, content = data["watermark"], data["content"]
    kw = dict(render_kwargs)
    kw.update(staged=False, bg_color=1, perturb=False, force_all_rays=True)
    if presum_adopt:   # the previous step's optimiser kernel left this message's pre-sum in the buffer (GraphedWatermarkLoop): no pass here
        model.adopt_presum(message)
    main = None
    if side_stream is not None and wm["rays_o_block"].is_cuda:
        main = torch.cuda.current_stream()
        # presum_first: the block render's samples were marched ahead -- nothing on the main stream to hide the pre-sum behind.  A sharded
        # codebook optimiser: the pre-sum ends in a collective (a segment boundary of a captured step), which must not be reached on
        # the forked side stream -- the capture would end with an unjoined stream.
        if presum_first or getattr(model, "codebook_shard", None) is not None:
            model.prepare_message(message)
        side_stream.wait_stream(main)
        # Issued first: the content render's field pass finds no pre-summed codebook yet and computes it right behind its own
        # march, on the side stream -- beside the block render's march, which does not need it; the block render's field pass
        # then waits for that event only (network._presum_consumed).
        if not blocks_first:
            with torch.cuda.stream(side_stream):
                content_pred_rgb = model.render(content["rays_o"], content["rays_d"], message, **kw)["image"]
            content_pred_rgb.record_stream(main)
    block_o, block_d, shard = local_blocks(wm)
    graphed = None
    eager_caller = block_graph is not None and main is None and shard is None and not presum_adopt
    # the content render's march and its one host read, in front of the block render: the read then waits for the march alone (NeRFRenderer.premarch)
    premarch = (eager_caller and content["rays_o"].is_cuda and model.training and torch.is_grad_enabled() and hasattr(model, "premarch") and getattr(model, "cuda_ray", False)
                and getattr(model, "point_capacity", None) is None and not torch.cuda.is_current_stream_capturing())
    if eager_caller and (distortion is None or isinstance(distortion, DistortionLayer)):
        whole = None
        if hasattr(block_graph, "usable_content") and content["images"].shape[-1] == 3:
            whole = (content["rays_o"], content["rays_d"], content["images"], float(lambda_w), float(lambda_i))
            if not block_graph.usable_content(whole, loss_w is loss_w_bce, color_space):
                whole = None
        if whole is None and premarch:       # (a StepGraph marches the content rays itself, into its own static record)
            model.premarch(content["rays_o"], content["rays_d"], kw.get("dt_gamma", 0), kw.get("max_steps", 1024))
            premarch = False
        graphed = block_graph.run(model, block_o, block_d, message, kw, distortion, whole)
        if graphed is not None and whole is not None:      # the whole step came out of the two captured graphs (blockgraph.StepGraph)
            _WatermarkLoss.last = None
            train_step.last_lambdas = (float(lambda_w), float(lambda_i))
            return graphed
    if premarch:
        model.premarch(content["rays_o"], content["rays_d"], kw.get("dt_gamma", 0), kw.get("max_steps", 1024))
    outputs = model.render(block_o, block_d, message, **kw) if graphed is None else None
    content_done = early_seed = None
    new_segment = False       # (a collective in front of the decoder ended the running capture segment: the side stream has to be forked again)
    if main is not None and blocks_first:
        # (the fork above is the content render's only parent; captured BEHIND the block render it is enqueued behind it: GraphedWatermarkLoop)
        with torch.cuda.stream(side_stream):
            content_pred_rgb = model.render(content["rays_o"], content["rays_d"], message, **kw)["image"]
            if content_backward_now is not None and color_space == "srgb" and content["images"].shape == content_pred_rgb.shape \
                    and content_pred_rgb.requires_grad:
                # content_backward_now = the scale of the image loss's seed (lambda_i x the data-parallel factor).  d(mean((c - gt)^2))/dc needs
                # nothing of the block render or the decoder, so the content render's backward follows its forward directly, on its stream: one chain
                # from the march to the scatter, no fork at the loss kernel (which then has ONE child, the decoder's backward: it stays on the
                # loss kernel's queue instead of hopping to another one, ~10 us of queue work on this runtime).  The loss kernel below still
                # computes the same MSE for the returned value, from the detached prediction; the seed is the same number it would leave.
                content_done = torch.cuda.Event()
                content_done.record(side_stream)      # what the loss kernel waits for: the forward, not the backward behind it
                early_seed = (content_pred_rgb.detach() - content["images"]) * (2.0 / content_pred_rgb.numel())
                if float(content_backward_now) != 1.0:
                    early_seed = early_seed * float(content_backward_now)
        content_pred_rgb.record_stream(main)
    if main is not None:
        fo.flush_plans()          # the block render's scatter plan: on the plan stream, behind the content render
    image = outputs["image"] if graphed is None else None
    if shard is not None:
        # the decoder's BatchNorm needs all D blocks (batch statistics, hidden_models.py:26): all-gather the rendered blocks, decode
        # them on every rank; the backward keeps this rank's rows.  Where the collective ends a captured segment both streams meet first
        # (a capture cannot end with a forked stream); captured inside the graph or issued eagerly it leaves the content render running
        # beside the decoder.
        if main is not None and dp.collective_ends_segment():
            main.wait_stream(side_stream)
            new_segment = True
        image = dp.gather_blocks(image, wm["rays_o_block"].shape[0], shard[0])
    if isinstance(distortion, str):
        distortion = DistortionLayer(distortion) if distortion != "none" else None
        if distortion is not None:
            distortion.draw(tuple(image.shape), image.device)
    fused_seed = keys_dev = None
    if graphed is not None:
        decoded, pred_rgb = graphed
    elif model.normalization is normalize_img and hasattr(model.msg_decoder, "decode_rendered"):
        from .hidden_models import _FusedDecoder
        _FusedDecoder.seed = None       # (only a seed written by THIS call's decoder pass may be used below)
        bce = None
        if loss_w is loss_w_bce and image.is_cuda and image.dim() == 4 and torch.is_grad_enabled():
            # d(lambda_w * mean BCE-with-logits(10 * decoded, message)) / d decoded is element-wise: the decoder's head kernel writes it (its backward
            # then starts right behind its forward: no loss kernel, no seed scaling between them)
            keys_dev = message.to(image.device, torch.float32).contiguous()
            if keys_dev.numel() == image.shape[0]:
                bce = (keys_dev, 10.0, float(lambda_w) * 10.0 / image.shape[0])
        decoded, pred_rgb = model.msg_decoder.decode_rendered(image, distortion, bce)    # clamp + distortion + permute + normalise inside layer 0
        fused_seed, _FusedDecoder.seed = _FusedDecoder.seed, None
    else:
        pred_rgb = torch.clamp(image, min=0, max=1)
        pred_rgb_dist = pred_rgb if distortion is None else distortion(pred_rgb, raw=image)
        decoded = model.msg_decoder(model.normalization(pred_rgb_dist.permute(0, 3, 1, 2)))
    if color_space == "linear":      # utils_wtmk_disen.py:603-604: converted IN PLACE, every step, as the reference does (its loader hands out fresh tensors)
        content["images"][..., :3] = srgb_to_linear(content["images"][..., :3])
    if content["images"].shape[-1] != 3:
        # C == 4: the reference itself cannot run this branch in the watermark stage -- `bg_color` is assigned only when C == 3 or bg_radius > 0
        # (utils_wtmk_disen.py:585-586) and read unconditionally at :590: UnboundLocalError
        raise NotImplementedError("RGBA ground truth: the reference's train_step raises UnboundLocalError here (bg_color is only assigned for 3-channel images, "
                                  "utils_wtmk_disen.py:585-590); blend the alpha channel into the images before the step")
    gt_rgb = content["images"]
    if main is not None and content_done is not None:
        # issued HERE, behind the decoder's forward launches: this runtime signals another queue only at the end of the run of launches a queue was
        # handed in one piece -- issued right behind the content render's forward, the loss kernel would wait for this backward as well
        if new_segment:
            side_stream.wait_stream(main)
        with torch.cuda.stream(side_stream):
            torch.autograd.backward([content_pred_rgb], [early_seed])
        content_pred_rgb = content_pred_rgb.detach()
        if not new_segment:
            # (new_segment: both streams met in front of the collective; the event belongs to the finished segment)
            main.wait_event(content_done)
    elif main is not None and not new_segment:
        # (new_segment: both streams already met in front of the collective and the side stream has not been forked into the new capture segment --
        #  waiting for it here would make the capturing stream depend on an event recorded outside the capture)
        main.wait_stream(side_stream)
    elif main is None:
        content_pred_rgb = model.render(content["rays_o"], content["rays_d"], message, **kw)["image"]
    keys = (keys_dev if keys_dev is not None and keys_dev.device == decoded.device else message.to(decoded.device)).unsqueeze(-1)
    if loss_w is loss_w_bce and decoded.is_cuda and all(t.dtype == torch.float32 for t in (content_pred_rgb, gt_rgb, decoded, keys)) \
            and gt_rgb.shape == content_pred_rgb.shape and keys.shape == decoded.shape:
        lossi, lossw, loss = _WatermarkLoss.apply(content_pred_rgb, gt_rgb, decoded, keys, float(lambda_w), float(lambda_i), 10.0)
        _WatermarkLoss.last = (loss, content_pred_rgb, decoded, *_WatermarkLoss.stash, float(lambda_w), float(lambda_i), fused_seed)
    else:
        lossi = ((content_pred_rgb - gt_rgb) ** 2).mean()
        lossw = loss_w(decoded, keys)
        loss = lambda_w * lossw + lambda_i * lossi
    train_step.last_lambdas = (float(lambda_w), float(lambda_i))
    return pred_rgb, gt_rgb, content_pred_rgb, lossi, lossw, loss


def reference_trainer_train_step(self, data, message):
    """Trainer.train_step of the reference (utils_wtmk_disen.py:579-646) as a method for ITS Trainer class: same arguments, same six return values
    (pred_rgb, gt_rgb, content_pred_rgb, lossi, lossw, loss), computed by this repo's train_step -- clamp, layout change, normalisation and the distortion
    layer inside the decoder's first launch, the three losses in one kernel -- instead of ~20 stock operators around model.render / model.msg_decoder.
    Read from `self` exactly what the reference's method reads: model, opt (splatted into render(), color_space), lambda_w, lambda_i, distortion,
    and opt.loss_w.  The drop-in directory's nerf/utils_wtmk_disen.py binds it (NERFSIG_DROPIN_OFF=train_step keeps the reference's own method)."""
    crit = getattr(self, "criterion", None)
    if crit is not None and not (isinstance(crit, torch.nn.MSELoss) and crit.reduction == "none"):
        # the fused step computes the reference's default content loss (criterion = MSELoss(reduction='none'), main_nerf_wtmk.py:108, applied at
        # utils_wtmk_disen.py:638); a Trainer built with another criterion trains through the reference's own method (the class this one shadows)
        for klass in type(self).__mro__[1:]:
            if "train_step" in vars(klass):
                return vars(klass)["train_step"](self, data, message)
        raise NotImplementedError(f"reference_trainer_train_step implements criterion = MSELoss(reduction='none'); got {crit!r} and no reference method to fall back to")
    images = data["watermark"].get("images")
    if images is not None and images.shape[-1] != 3 and not self.model.bg_radius > 0:
        # the reference assigns bg_color only for 3-channel block images (:585-586) and reads it unconditionally (:590)
        raise UnboundLocalError("local variable 'bg_color' referenced before assignment (utils_wtmk_disen.py:590: the watermark stage needs 3-channel block images)")
    kind = getattr(self, "distortion", "none") or "none"
    layer = None
    if kind != "none":
        layer = self.__dict__.get("_nsig_distortion_layer")
        if layer is None or layer.name != kind:
            layer = self.__dict__["_nsig_distortion_layer"] = DistortionLayer(kind)
        o = data["watermark"]["rays_o_block"]
        layer.draw(tuple(o.shape), o.device)
    name = getattr(self.opt, "loss_w", "bce")
    if name not in ("bce", "mse"):
        raise NotImplementedError
    graph = self.__dict__.get("_nsig_block_graph")
    if graph is None:
        from .blockgraph import StepGraph
        graph = self.__dict__["_nsig_block_graph"] = StepGraph()
    return train_step(self.model, data, message, vars(self.opt), lambda_w=self.lambda_w, lambda_i=self.lambda_i, loss_w=loss_w_bce if name == "bce" else loss_w_mse,
                      color_space=getattr(self.opt, "color_space", "srgb"), distortion=layer, block_graph=graph)


def eval_step(model, data, message, render_kwargs, render_whole=True, lambda_w=1.0, lambda_i=1.0, loss_w=loss_w_bce, color_space="srgb", distortion=None):
    """Trainer.eval_step (utils_wtmk_disen.py:648-702) for 3-channel images / srgb.  distortion: as in train_step -- the reference applies its
    distortion layer to the evaluated blocks as well (:666), with a fresh draw per call.
    render_whole=False: the watermark blocks rendered with the message, decoded, BCE against the message (+ MSE against
    data['images_block'] when given) -- what test_bitacc evaluates; render_whole=True: the full view staged in max_ray_batch chunks
    (`render(staged=True)`) against data['images'] -- what test_image evaluates.  The reference never calls model.eval() on these
    paths (:832, :951), so whether the training- or the eval-mode kernels run is the caller's `model.train()` / `model.eval()`.
    Returns (pred_rgb, pred_depth, gt_rgb, decoded_message, lossi, lossw, loss) like the reference."""
    dev = message.device
    decoded, gt_rgb = None, None
    lossi, lossw, loss = (torch.zeros(1, device=dev) for _ in range(3))
    kw = dict(render_kwargs)
    if not render_whole:
        out = model.render(data["rays_o_block"], data["rays_d_block"], message, staged=False, bg_color=1, perturb=False, force_all_rays=True, **kw)
        pred_rgb, pred_depth = torch.clamp(out["image"], min=0, max=1), out["depth"]
        if isinstance(distortion, str):
            distortion = DistortionLayer(distortion) if distortion != "none" else None
        pred_rgb_dist = pred_rgb
        if distortion is not None:
            distortion.draw(tuple(pred_rgb.shape), pred_rgb.device)
            pred_rgb_dist = distortion(pred_rgb)
        decoded = model.msg_decoder(model.normalization(pred_rgb_dist.permute(0, 3, 1, 2)))
        gt_rgb = data.get("images_block")
        if gt_rgb is not None:
            lossi = ((pred_rgb - gt_rgb) ** 2).mean()
        lossw = loss_w(decoded, message.to(decoded.device).unsqueeze(-1))
        loss = lambda_w * lossw + lambda_i * lossi
    else:
        B, H, W, C = data["images"].shape
        images = data["images"]
        if color_space == "linear":      # :691-692, in place like the reference
            images[..., :3] = srgb_to_linear(images[..., :3])
        # evaluation uses a fixed white background (:695-699)
        gt_rgb = images[..., :3] * images[..., 3:] + 1 * (1 - images[..., 3:]) if C == 4 else images
        out = model.render(data["rays_o"], data["rays_d"], message, staged=True, bg_color=1, perturb=False, force_all_rays=True, **kw)
        pred_rgb = torch.clamp(out["image"].reshape(-1, H, W, 3), min=0, max=1)
        pred_depth = out["depth"].reshape(-1, H, W)
    return pred_rgb, pred_depth, gt_rgb, decoded, lossi, lossw, loss


def test_step(model, data, message, render_kwargs, bg_color=None, perturb=False):
    """Trainer.test_step (utils_wtmk_disen.py:704-722): a staged full-view render, clamped.  Returns (pred_rgb [B,H,W,3], pred_depth [B,H,W])."""
    H, W = data["H"], data["W"]
    out = model.render(data["rays_o"], data["rays_d"], message, staged=True, bg_color=bg_color, perturb=perturb, **dict(render_kwargs))
    return torch.clamp(out["image"].reshape(-1, H, W, 3), min=0, max=1), out["depth"].reshape(-1, H, W)


test_step.__test__ = False      # (not a pytest test, whatever its name)


class WatermarkLoop:
    """Loop body of train_one_epoch (utils_wtmk_disen.py:1164-1181) for one model replica."""

    def __init__(self, model, optimizer, render_kwargs, lambda_w=1.0, lambda_i=1.0, lr_scheduler=None, use_sink=True, side_stream=None, distortion="none",
                 distortion_seed=0):
        self.distortion = None if distortion in (None, "none") else (distortion if isinstance(distortion, DistortionLayer) else DistortionLayer(distortion, distortion_seed))
        self.side_stream = side_stream
        self.plan_stream = side_stream   # scatter plans queue behind the content render
        self.model, self.optimizer, self.lr_scheduler = model, optimizer, lr_scheduler
        self.render_kwargs = dict(render_kwargs)
        self.lambda_w, self.lambda_i = lambda_w, lambda_i
        dev = next(model.parameters()).device
        self.sink = fo.GradSink(dev) if use_sink else None
        model.grad_sink = self.sink
        self.exchange = GradExchange(list(model.msg_decoder.parameters()))
        self.last = None

    def _sharded(self, data):
        return local_blocks(data["watermark"])[2] is not None

    def step(self, data, message):
        """message: float tensor of 0./1. (keep it on the CPU to avoid the D2H read of its bits)."""
        if getattr(self.model, "device_select", False):
            raise RuntimeError("the model is in device-select mode (a GraphedWatermarkLoop prepared it): its sink records all 2D tables as "
                               "selected, which the eager optimiser step would update wrongly; call GraphedWatermarkLoop.close() first")
        self.optimizer.zero_grad(set_to_none=True)
        if self.sink is not None:
            self.sink.zero_()
        if self.distortion is not None:      # this step's random draws (all D blocks reach the decoder, sharded or not)
            o = data["watermark"]["rays_o_block"]
            self.distortion.draw(tuple(o.shape), o.device)
        prev = fo.set_plan_stream(self.plan_stream)    # the scatter plans of both renders leave the critical path too
        try:
            out = train_step(self.model, data, message, self.render_kwargs, self.lambda_w, self.lambda_i, side_stream=self.side_stream, distortion=self.distortion)
        finally:
            fo.set_plan_stream(prev)
        sharded = self._sharded(data)
        backward_from_loss_kernel(out, dp.content_grad_scale(sharded))
        if self.side_stream is not None:
            torch.cuda.current_stream().wait_stream(self.side_stream)
        if self.sink is None and exchange_active():
            raise RuntimeError("data-parallel steps need the GradSink route (use_sink=True): per-table gradients are not exchanged")
        self.exchange.shared_scale = 1.0 if sharded else None      # sharded blocks: G is a plain sum (content seeds carry 1/world)
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


class GraphedWatermarkLoop:
    """The same loop body as WatermarkLoop, captured once into a hipGraph and replayed.

    What makes the step capturable (LABNOTES.md section 8): the point buffers of both renders have a fixed capacity, so
    the march never reads a count back (`march_rays_train_capacity`); the message lives in a device tensor and the
    pre-sum / Adam kernels select their tables on the device (`hg_codebook_presum_sel`, `opt_codebook_adam_sel`), so no
    launch argument depends on the message; Adam step counts and the learning rate are device scalars.  Per step the
    host only copies D message floats, one learning rate and (optionally) new rays into static tensors and replays.
    With more than one rank the gradient exchange runs between two graphs (forward+backward | optimiser).

    Capacity: `prepare()` runs one ordinary (synchronising) step to learn the padded point counts of the two renders and
    adds `headroom`; `overflowed()` reports (one host read) whether any replay since the last check produced more
    points than that -- such a step dropped the rays that did not fit, like the reference's bounded mode."""

    def __init__(self, model, optimizer, render_kwargs, data, lambda_w=1.0, lambda_i=1.0, lr_lambda=None, headroom=0.0, native_dense_adam=True,
                 overlap_content=True, march_ahead=None, presum_in_adam=True, stage_in_graph=True, content_headroom=None,
                 content_sampler=None, fixed_blocks=False, distortion="none", distortion_seed=0):
        """distortion: all five of the reference's kinds run inside the captured step -- noise | brightness | blurring as part of the decoder's first launch,
        rotation | scaling as one resampling launch in front of it.  Their random parameters are re-drawn on the device every replay (wm_distort_draw, keyed
        by distortion_seed and the replay count), except the scaling factor: it sets the decoder's input WIDTH, so prepare() captures the step once per
        possible width and step() draws the factor on the host (distortion.host_uniform, same key) and replays the capture of that width.
        presum_in_adam: the captured optimiser kernel also writes the pre-summed codebook of the NEXT step's message
        (opt_codebook_adam_sel_next: +9 % traffic inside an HBM-streaming kernel instead of a 128 MiB pass at the head of every step).
        The next message is handed over one step early -- `step(message_k, next_message=message_k1)`, a one-element look-ahead over
        the random draws of utils_wtmk_disen.py:1165; a step whose message was not announced runs the stand-alone pre-sum before
        its replay, so `step(message)` alone stays correct.
        march_ahead (default: with overlap_content): the block render's samples are marched at the end of the previous
        replay, beside the optimiser (the march needs rays and occupancy grid only -- nothing a step updates -- and the
        codebook Adam is an HBM stream that leaves the ALUs idle: 166 us together against 224 us one after the other,
        tools/overlap_probe.py).  The block rays of the NEXT step therefore have to be in the static buffers when a replay
        starts: pass the next step's data as `step(..., next_data=...)` (its content part is applied at that step);
        `step(..., data=...)` still works -- it re-marches before the replay, un-overlapped.  The content render's march stays
        at the head of its own step on the side stream: next to the optimiser as well, the two marches took longer than the
        optimiser and the pre-sum lost its cover."""
        self.distortion = None if distortion in (None, "none") else (distortion if isinstance(distortion, DistortionLayer) else DistortionLayer(distortion, distortion_seed))
        self.march_ahead = overlap_content if march_ahead is None else bool(march_ahead)
        # fixed_blocks (off by default): the watermark-block rays are one pair of tensors per dataset
        # (nerf/provider_wtmk.py:442-494) and everything their field pass reads except the codebook is frozen in this stage, so the
        # loop marches them ONCE, keeps the base-level feature planes and the scatter plan (NeRFNetwork.fix_rays / fieldops.FixedPoints)
        # and a step only gathers the codebook level for them -- same kernels' results, bit-identical training
        # (tests/test_gpu_fixed.py).  New block rays (`data` / `next_data` with a "watermark" part), a loaded
        # checkpoint (invalidate) or a re-sized capture refresh the kept buffers in place before the next replay.
        self.fixed_blocks = bool(fixed_blocks)
        self._refix_pending = False
        self._kept_key = None
        # content_sampler (rays.DeviceRaySampler): the step draws its own content batch -- pose, pixels, rays, ground truth -- on the
        # device, inside the captured graph, from the replay count; `data` / `next_data` then carry no content part
        self.content_sampler = content_sampler
        self.presum_in_adam = bool(presum_in_adam)
        # stage_in_graph: the captured step opens with loop_step_begin, which zero-fills G and fetches the step's message words from the
        # pinned ring itself (slot = replays so far, counted on the device) -- no host-to-device copy command between two replays
        self.stage_in_graph = bool(stage_in_graph)
        self._s_for = None            # host copy of the message the pre-sum buffer currently belongs to (None: unknown / stale)
        self.marched = None
        self._pending_content = None
        self.native_dense_adam = native_dense_adam
        self.side_stream = torch.cuda.Stream() if overlap_content else None
        self.plan_stream = self.side_stream   # scatter plans queue behind the content render
        # The decoder's parameter-gradient kernels (~85 us, needed only by the optimiser) leave the main stream, so that the block render's backward waits for
        # the image gradient alone: with a side stream they get a stream of their own, the content render's backward is issued first
        # (backward_from_loss_kernel) -- it runs beside the decoder's backward chain -- and the block render is captured first (_blocks_issued_first).
        # The alternative (everything at the step's tail on the content render's stream) was measured slower at every world size
        # (profiles/r03_backward_schedule.txt) and is gone.
        self.weights_stream = self.side_stream
        self.content_backward_first = False
        if not hasattr(optimizer, "step_shared_sel"):
            raise TypeError("GraphedWatermarkLoop needs nerf_signature_amd.optim.CodebookAdam(capturable=True)")
        if not any(t.requires_grad for t in model.msg_encoder.tables()):
            raise NotImplementedError("GraphedWatermarkLoop captures the step that trains the codebook; with a frozen codebook "
                                      "(finetune_decoder=True) drive the model with WatermarkLoop")
        self.model, self.optimizer = model, optimizer
        self.render_kwargs = dict(render_kwargs)
        self.lambda_w, self.lambda_i = lambda_w, lambda_i
        self.lr_lambda, self.headroom = lr_lambda, headroom
        # content rays change every step (a new pose / new pixels, utils_wtmk_disen.py:1164), so their sample total varies: own headroom
        self.content_headroom = headroom if content_headroom is None else content_headroom
        dev = next(model.parameters()).device
        self.device = dev
        self._warm_sink = torch.zeros(1, dtype=torch.float32, device=dev)      # (hg_warm_tables: never written)
        D = model.message_dim
        # G and (behind it) room for the decoder's flat gradient block in one allocation: one all-reduce per step (dp.GradExchange)
        self.sink = fo.GradSink(dev, tail=sum(p.numel() for p in model.msg_decoder.parameters()))
        model.grad_sink = self.sink
        self.exchange = GradExchange(list(model.msg_decoder.parameters()), average=not native_dense_adam)
        self.data = {"watermark": {k: v.clone() for k, v in data["watermark"].items()},
                     "content": {k: v.clone() for k, v in data["content"].items()}}
        # this step's message, then the next step's, then this step's learning rate: what the host hands a step, staged together -- with an lr schedule the
        # step's opening kernel fetches the rate with the message (a `fill_` of its own between two replays was a 4 us launch on the critical path)
        self.msg_all = torch.zeros(2 * D + 1, dtype=torch.float32, device=dev)
        self.msg_dev, self.msg_next_dev = self.msg_all[:D], self.msg_all[D:2 * D]
        # The host runs ahead of the GPU by many replays, so the pinned staging buffer of a step must not be rewritten until its
        # asynchronous copy has executed: a ring of buffers, each guarded by an event.
        # words a step stages (= the ring's row width): without a schedule the rate is whatever the tensor holds (a caller may write it) and is not staged
        self.stage_width = 2 * D + (1 if lr_lambda is not None else 0)
        self.msg_ring = torch.zeros(16, self.stage_width, dtype=torch.float32).pin_memory()      # one row per in-flight step
        self.msg_events = [None] * len(self.msg_ring)
        self.stage_counter = torch.zeros(1, dtype=torch.int32, device=dev)           # replays so far (advanced by loop_step_begin)
        self.ring_dev = nv.fn("nsig_host_device_pointer")(ctypes.c_void_p(self.msg_ring.data_ptr())) if self.stage_in_graph else None
        if self.stage_in_graph and not self.ring_dev:
            self.stage_in_graph = False         # the ring is not device-mapped on this platform: keep the per-step copy
        self.base_lr = float(optimizer.param_groups[0]["lr"])
        self.lr_dev = self.msg_all[2 * D]        # (a 0-dim view: the optimiser's tensor lr)
        self.lr_dev.fill_(self.base_lr)
        for g in optimizer.param_groups:
            g["lr"] = self.lr_dev          # tensor lr: the captured optimiser reads it on the device
        self.tables = model.msg_encoder.tables()
        self.graphs = None
        self.segments, self.between = [], []
        self.sharded = False
        self.opt_shard = None
        self.out = None
        self.steps_done = 0
        self._replays = 0             # replays of the opening graph so far == the device's stage_counter (ring slot of the next step)
        self.capacity_rows = None
        if not hasattr(model, "_graphed_loops"):
            model._graphed_loops = []
        model._graphed_loops.append(self)

    def resume_at(self, step):
        """Continue a run at iteration `step` (before the first replay): the learning-rate schedule, the device-side loader's batch
        sequence and the message ring's slot all count from there."""
        if self.graphs is not None and self._replays:
            raise RuntimeError("resume_at: the loop has already replayed steps")
        self.steps_done = int(step)
        self._replays = int(step)
        self.stage_counter.fill_(int(step))

    def invalidate(self):
        """The model's parameters were overwritten from outside (checkpoint.load_checkpoint): the pre-sum buffer no longer belongs to
        any announced message, so the next step runs the stand-alone pre-sum before its replay; the packed MLP weights are refreshed
        here, in the buffer the graph reads (the captured step never re-packs)."""
        self._s_for = None
        self.model._packed()
        if self.fixed_blocks and self.graphs is not None:
            self._fix_blocks()       # the base tables may have been overwritten: the kept feature planes, in place

    @torch.no_grad()
    def gather_codebook(self, optimizer_state=True):
        """With the optimiser sharded over the ranks (dp.optimizer_shard) a rank's copies of the tables it does not own are stale.  Brings
        every rank's codebook (and, optionally, Adam's moments and step counts of those tables) up to date with one all-gather per kind:
        call before saving a checkpoint, evaluating with a message through the host-side selection, or leaving the captured loop."""
        if self.opt_shard is None:
            return
        import torch.distributed as dist
        from .optim import _prepare_device_state
        b0, b1 = self.opt_shard
        world = dist.get_world_size()
        _prepare_device_state(self.optimizer, self.tables)      # (state of tables this rank never stepped: zeros, about to be overwritten)
        kinds = [lambda t: t.data]
        if optimizer_state:
            dp_state = self.optimizer.state
            kinds += [lambda t, k=k: dp_state[t][k] for k in ("exp_avg", "exp_avg_sq", "step") if all(k in dp_state[t] for t in self.tables)]
        for get in kinds:
            own = torch.stack([get(t).reshape(-1) for t in self.tables[2 * b0:2 * b1]])          # [2 * bits per rank, n]
            every = torch.empty((world,) + tuple(own.shape), dtype=own.dtype, device=own.device)
            dp._all_gather_into(every.view(world * own.shape[0], -1), own)
            for i, t in enumerate(self.tables):
                get(t).reshape(-1).copy_(every.view(len(self.tables), -1)[i])
        from .optim import _bump_versions
        _bump_versions(self.tables)
        self.model._codebook_stale = False

    def close(self):
        """Detach from the model: the eager loop (or another graphed loop) may drive it again."""
        self.gather_codebook()
        self.model.codebook_shard = None
        self.model.device_select = False
        self.model.point_capacity = None
        if self in getattr(self.model, "_graphed_loops", ()):
            self.model._graphed_loops.remove(self)

    # -- pieces of one step (executed eagerly during warm-up, then under capture)
    def _forward_backward(self):
        if self.stage_in_graph and torch.cuda.is_current_stream_capturing():
            nv.call("loop_step_begin", nv.ptr(self.sink.G), self.sink.G.numel(), ctypes.c_void_p(self.ring_dev), len(self.msg_ring),
                    self.stage_width, nv.ptr(self.stage_counter), nv.ptr(self.msg_all), nv.stream())
        else:
            self.sink.zero_()
            if self.content_sampler is not None and torch.cuda.is_current_stream_capturing():
                self.stage_counter += 1      # (the opening kernel counts the replays when it stages the message)
        if self.content_sampler is not None:
            ct = self.data["content"]
            self.content_sampler.sample_into(self.stage_counter, ct["rays_o"], ct["rays_d"], ct["images"])
        if self.distortion is not None and self.distortion.name != "scaling":      # this step's draws, from (seed, replay count): the same on every rank
            o = self.data["watermark"]["rays_o_block"]
            self.distortion.draw_on_device(self.stage_counter, tuple(o.shape), o.device)
        prev = fo.set_plan_stream(self.plan_stream)    # the scatter plans need the sample positions only: beside the forward pass
        try:
            out = train_step(self.model, self.data, self.msg_dev, self.render_kwargs, self.lambda_w, self.lambda_i, side_stream=self.side_stream,
                             presum_first=self.marched is not None,
                             presum_adopt=self.presum_in_adam and torch.cuda.is_current_stream_capturing(),
                             blocks_first=self.side_stream is not None and self._blocks_issued_first(),
                             content_backward_now=(self.lambda_i * dp.content_grad_scale(self.sharded)) if (self.content_backward_first and self.side_stream is not None) else None,
                             distortion=self.distortion)
        finally:
            fo.set_plan_stream(prev)
        set_weights_stream(self.weights_stream)
        set_grad_arena(self.sink.tail)
        try:
            backward_from_loss_kernel(out, dp.content_grad_scale(self.sharded), self.side_stream, self.content_backward_first)
        finally:
            set_weights_stream(None)
            set_grad_arena(None)
        # the content render's backward ends in a side effect (the shared gradient): join it explicitly; likewise the decoder's parameter gradients
        join_stream(torch.cuda.current_stream(), self.side_stream)
        if self.weights_stream is not self.side_stream:
            join_stream(torch.cuda.current_stream(), self.weights_stream)
        return out

    def _optimise(self, defer_collective=False):
        """The optimiser step.  defer_collective: return the sharded optimiser's closing all-reduce as a callable instead of running it
        (the caller runs it once every forked stream has joined: a collective may end a captured segment)."""
        post = None
        scale = 1.0 / world_size() if self.native_dense_adam else 1.0    # the exchange leaves sums: the mean is taken here
        # sharded blocks: G is already sum_r G_block_r + mean_r G_content_r (the content seeds carried 1/world): no factor
        scale_cb = 1.0 if (self.sharded and self.native_dense_adam) else scale
        tables, msg, msg_next = self.tables, self.msg_dev, self.msg_next_dev
        if self.opt_shard is not None:      # ZeRO-style: this rank updates the tables of its own bits only (dp.optimizer_shard)
            b0, b1 = self.opt_shard
            tables, msg, msg_next = self.tables[2 * b0:2 * b1], self.msg_dev[b0:b1], self.msg_next_dev[b0:b1]
        if self.presum_in_adam:     # ... and the next step's pre-sum, in place (both renders of this step are done with the buffer)
            S = self.model._presum_cache[1]
            self.optimizer.step_shared_sel(tables, msg, self.sink.G, self.lr_dev, scale_cb, next_message_dev=msg_next, S_next=S)
            if self.opt_shard is not None:  # the ranks' partial pre-sums of the next message add up to the whole one
                import torch.distributed as dist
                post = lambda: dp.collective(lambda: dist.all_reduce(S, op=dist.ReduceOp.SUM), last=True, name="all_reduce_presum")
        else:
            self.optimizer.step_shared_sel(tables, msg, self.sink.G, self.lr_dev, scale_cb)
        if self.native_dense_adam:
            self.optimizer.step_dense(self.lr_dev, scale)      # the decoder's parameters: opt_adam_dense
        else:
            self.optimizer.step()
        if post is not None and not defer_collective:
            post()
            post = None
        return post

    def _march_ahead(self):
        kw, wm = self.render_kwargs, self.data["watermark"]
        args = (kw.get("dt_gamma", 0), kw.get("max_steps", 1024))   # (march_ahead(..., phase="count" | "write") could split the walk from the writes: measured, slower)
        # the block render's march only: the content render's stays at the head of its step, on the side stream, where it
        # overlaps the pre-sum and the block encoder (both marches next to the optimiser took longer than the optimiser)
        block_o, block_d, _ = local_blocks(wm)       # (this rank's shard of the blocks when they are split over the ranks)
        if self.fixed_blocks:                        # marched once, outside the step (_fix_blocks)
            self.marched = self.marched[:1]
        else:
            self.marched = (self.model.march_ahead(block_o, block_d, *args),)

    @torch.no_grad()
    def _fix_blocks(self):
        """(Re-)march this rank's block rays and (re-)compute what their field pass keeps across steps, in place (eager, between replays)."""
        kw = self.render_kwargs
        block_o, block_d, _ = local_blocks(self.data["watermark"])
        rec = self.model.fix_rays(block_o, block_d, kw.get("dt_gamma", 0), kw.get("max_steps", 1024))
        self.marched = (rec,) + tuple(self.marched[1:] if self.marched else ())
        self._refix_pending = False
        self._kept_key = self._kept_inputs_key()

    def _optimise_and_march(self):
        """The optimiser step and, beside it on the side stream, the march of the next step's samples."""
        if not self.march_ahead or self.fixed_blocks:
            return self._optimise()
        main = torch.cuda.current_stream()
        if self.side_stream is not None:
            self.side_stream.wait_stream(main)      # both backward passes of this step are done with the buffers
            with torch.cuda.stream(self.side_stream):
                self._march_ahead()
                self._warm_tables()
        post = self._optimise(defer_collective=True)
        if self.side_stream is not None:
            main.wait_stream(self.side_stream)
        else:
            self._march_ahead()
            self._warm_tables()
        if post is not None:
            post()                                  # (a collective: only after the streams have joined)

    def _warm_tables(self):
        """The frozen base tables read once, each through the XCD that will gather from it (hg_warm_tables), behind the march of the next step's
        block samples and beside the optimiser: the next step's block encoder -- the step's longest kernel, at its head -- otherwise starts on
        caches the optimiser's 836 MiB stream has flushed (282-290 us; 258-270 us behind this pass, which takes ~20 us alone).  Same-box A/B of
        the bench step: 1.026-1.031 -> 1.005-1.019 ms (profiles/r03_warm_tables.txt).  Only where the optimiser outlasts it: with the codebook
        optimiser sharded over >= 4 ranks (a 22 us pass over D/R tables) the warm-up would end the step (emulated rank of 4 / 8: +1.5 / +2 %).
        Hence off there."""
        if self.opt_shard is not None or self.fixed_blocks or getattr(self.model, "_presum_cache", None) is None:
            return
        tables = nv.ptr_array([t.detach() for t in self.model.encoder.tables()])
        nv.call("hg_warm_tables", tables, nv.ptr(self.model._presum_cache[1]), nv.ptr(self._warm_sink), nv.stream())

    def point_counts(self):
        """(block, content) sample totals of the last step (one host read)."""
        if self.marched is not None:      # the block render has its own counter, the content render the ring's latest row
            return int(self.marched[0]["counter"][0]), int(self.model.step_counter[self.capacity_rows[-1], 0])
        a, b = self.model.step_counter[self.capacity_rows, 0].tolist()
        return (a, b) if self._blocks_issued_first() else (b, a)     # issue order: with a side stream the content render comes first

    def _blocks_issued_first(self):
        """Which render train_step issues (and a capture records) first: the block render ("beside" schedule) or the content render, on its side
        stream ("tail").  With its default packet capture this runtime starts a node whose parent sits on another of
        the graph's internal streams only when that stream has finished the whole run of nodes it was handed in one piece (tools/graph_dot.py,
        profiles/r03_graph_capture_order.txt): captured behind the content render's nine short kernels, the block render -- and the all-gather and
        the decoder behind it -- started when all nine were done; captured first, it is the content render that waits (for the decoder's forward
        chain, beside whose backward chain it then runs): emulated rank of 2 / 4 / 8: 0.846 -> 0.828, 0.610 -> 0.605, 0.534 -> 0.527 ms; one rank
        (where the wait keeps the content render's encoder away from the block render's): 1.050-1.068 -> 1.025-1.045 ms."""
        return self.side_stream is None or self.content_backward_first

    def _kept_inputs_key(self):
        """Versions of everything the kept planes / marched samples were computed from (in-place writes through torch bump them)."""
        m = self.model
        return m.grid_key() + tuple((t.data_ptr(), t._version) for t in m.encoder.tables())

    def _set_inputs(self, message, data, next_data=None, next_message=None, eager_copy=True):
        if self.fixed_blocks and self.graphs is not None and not self._refix_pending and self._kept_key != self._kept_inputs_key():
            self._refix_pending = True      # a base table or the occupancy grid was written since the blocks were fixed (17 attribute reads per step)
        if self._refix_pending:      # the previous call's next_data brought new block rays: they are this step's now
            self._fix_blocks()
        slot = self._replays % len(self.msg_ring)        # == the device's replay count modulo the ring (advanced right behind every replay of g1)
        if self.msg_events[slot] is not None:
            self.msg_events[slot].synchronize()      # blocks only if the GPU is a whole ring behind
        D = self.msg_dev.numel()
        self.msg_ring[slot][:D].copy_(message.detach().to("cpu", torch.float32))
        # an unannounced next message: the optimiser pre-sums for this step's bits again (harmless) and _s_for goes stale
        self.msg_ring[slot][D:2 * D].copy_((message if next_message is None else next_message).detach().to("cpu", torch.float32))
        if self.lr_lambda is not None:      # this step's learning rate travels with its message
            self.msg_ring[slot][2 * D] = self.base_lr * self.lr_lambda(self.steps_done)
        if eager_copy:      # (a replay with stage_in_graph fetches the row itself; step() then guards the slot with an event behind the replay)
            self.msg_all[:self.stage_width].copy_(self.msg_ring[slot][:self.stage_width], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            self.msg_events[slot] = ev
        if self._pending_content is not None:     # the content part that came with the block rays marched at the end of the last replay
            for k, v in self._pending_content.items():
                self.data["content"][k].copy_(v, non_blocking=True)
            self._pending_content = None
        if data is not None:           # either part may be omitted: the watermark pose is fixed per dataset, content rays change every step
            for part in ("watermark", "content"):
                for k, v in data.get(part, {}).items():
                    self.data[part][k].copy_(v, non_blocking=True)
            if self.fixed_blocks and self.graphs is not None and "watermark" in data:
                self._fix_blocks()              # new block rays: marched and their kept planes refreshed before the replay
            if self.march_ahead and self.graphs is not None and "watermark" in data and not self.fixed_blocks:
                self._march_ahead()             # this step's rays arrived only now: march them before the replay
        if next_data is not None:
            if not self.march_ahead:
                raise ValueError("next_data needs march_ahead=True")
            for k, v in next_data.get("watermark", {}).items():     # nothing in the replay reads the block rays before its closing march
                self.data["watermark"][k].copy_(v, non_blocking=True)
                self._refix_pending = self.fixed_blocks             # (no closing march with fixed blocks: re-fixed ahead of the next replay)
            content = next_data.get("content")
            if content is not None:
                self._pending_content = content              # marched (and compared with its images) inside its own step

    @torch.no_grad()
    def _snapshot(self):
        params = [p for g in self.optimizer.param_groups for p in g["params"]]
        state = {p: {k: (v.clone() if torch.is_tensor(v) else v) for k, v in self.optimizer.state[p].items()} for p in params if len(self.optimizer.state[p])}
        return [(p, p.detach().clone()) for p in params if p.requires_grad], state

    @torch.no_grad()
    def _restore(self, snapshot):
        saved_params, saved_state = snapshot
        for p, v in saved_params:
            p.copy_(v)
        for p, st in self.optimizer.state.items():
            for k, v in st.items():
                if not torch.is_tensor(v):
                    continue
                if p in saved_state and k in saved_state[p]:
                    v.copy_(saved_state[p][k].to(v.device))
                else:
                    v.zero_()      # state created by the warm-up: back to its initial value, same storage (the graph holds its address)

    def prepare(self, message):
        """One synchronising step to size the point buffers, warm-up on a side stream, then capture."""
        model = self.model
        model.device_select = False
        model.point_capacity = None
        self.optimizer.zero_grad(set_to_none=True)
        self.sink.zero_()
        block_o, block_d, shard = local_blocks(self.data["watermark"])
        self.sharded = shard is not None
        self.exchange.shared_scale = 1.0 if self.sharded else None
        self.opt_shard = dp.optimizer_shard(model.message_dim) if self.sharded else None
        model.codebook_shard = self.opt_shard
        if self.distortion is not None and self.distortion.name == "scaling":      # (its one parameter is a host value: the buffer the kernels read it from)
            o = self.data["watermark"]["rays_o_block"]
            self.distortion._buffers(tuple(o.shape), o.device)
            self.distortion.set_scaling(host_uniform(self.distortion.seed, self._replays, 0.75, 1.25))
        beside = self.side_stream is not None
        self.content_backward_first = beside
        self.weights_stream = self.side_stream if not beside else (getattr(self, "_own_weights_stream", None) or torch.cuda.Stream())
        self._own_weights_stream = self.weights_stream if beside else None
        with torch.no_grad():   # sizes only: the two renders of a step, in order (block, content)
            model.render(block_o, block_d, message, staged=False, bg_color=1,
                         perturb=False, force_all_rays=True, **self.render_kwargs)
            n_block = int(model.step_counter[(model.local_step - 1) % 16, 0])
            model.render(self.data["content"]["rays_o"], self.data["content"]["rays_d"], message, staged=False, bg_color=1, perturb=False,
                         force_all_rays=True, **self.render_kwargs)
            n_content = int(model.step_counter[(model.local_step - 1) % 16, 0])
        from .raymarching import padded_point_count
        rays_block = block_o.numel() // 3
        rays_content = self.data["content"]["rays_o"].numel() // 3
        if rays_block == rays_content:
            raise ValueError("block and content renders must have different ray counts to carry separate capacities")
        cap_block = padded_point_count(int(n_block * (1.0 + self.headroom)))
        cap_content = padded_point_count(int(n_content * (1.0 + self.content_headroom)))
        model.point_capacity = {rays_block: cap_block, rays_content: cap_content}
        model.device_select = True
        self._set_inputs(message, None)

        # Warm-up on a side stream (library handles, lazily created optimiser state, MIOpen algorithm choice must all
        # exist before capture).  The warm-up iterations must not train: parameters and optimiser state are restored.
        snapshot = self._snapshot()
        if self.fixed_blocks:
            if not self.march_ahead:
                raise ValueError("fixed_blocks needs march_ahead=True (the kept samples live in the march-ahead record)")
            self._fix_blocks()       # before the warm-up, so that it runs -- and loads -- the kernels the captured step will use
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                self.optimizer.zero_grad(set_to_none=True)
                self._forward_backward()
                self.exchange(self.sink.G)
                self._optimise()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self._restore(snapshot)

        # One capture per shape the step can take: a single one, except under `distortion='scaling'`, whose factor sets the decoder's input width
        # (floor(W * sf), sf in [0.75, 1.25)): one capture per width, each in a memory pool of its own; step() draws the factor on the host
        # (distortion.host_uniform: a function of the seed and the step index) and replays the capture of its width.
        self.variants = {}
        if self.distortion is not None and self.distortion.name == "scaling":
            W = self.data["watermark"]["rays_o_block"].shape[2]
            a_factor_of = {scaled_width(W, f): f for f in (0.75 + 0.5 * i / 4096.0 for i in range(4096))}
            for w in sorted(a_factor_of):
                self.distortion.set_scaling(a_factor_of[w])
                self.variants[w] = self._capture(cap_block, cap_content)
        else:
            self.variants[None] = self._capture(cap_block, cap_content)
        self._select(next(iter(self.variants)))
        return self

    def _select(self, key):
        """Make the capture of this shape the one step() replays and overflowed() / point_counts() read."""
        v = self.variants[key]
        self.segments, self.between, self.graphs, self.out = v["segments"], v["between"], v["graphs"], v["out"]
        self.capacity_rows, self.capacities = v["capacity_rows"], v["capacities"]

    def _capture(self, cap_block, cap_content):
        model = self.model
        self.optimizer.zero_grad(set_to_none=True)
        if self.march_ahead and not self.fixed_blocks:
            self._march_ahead()      # the first replay's samples (buffers outside the graph's pool, re-marched in place by every replay)
        # Segmented capture: one hipGraph per stretch between collectives.  A collective reached while capturing (dp.collective: the
        # all-gather of the rendered blocks, the gradient all-reduce) ends the running capture, is remembered as the eager call that
        # follows that segment in every replay, and the next segment begins in the same memory pool.  One rank: a single segment.
        # thread_local: with more than one rank the process group's watchdog thread queries events while we capture; only this
        # thread's calls belong to the capture.
        self.segments, self.between = [torch.cuda.CUDAGraph()], []

        open_capture = [True]

        def boundary(fn, last=False):
            self.segments[-1].capture_end()
            fo.forget_plan_events()          # events recorded in the finished capture must not be waited on in the next one
            self.model._presum_event = None
            self.between.append(fn)
            if last:                         # the step ends with this collective: no empty segment behind it
                open_capture[0] = False
                return
            g = torch.cuda.CUDAGraph()
            self.segments.append(g)
            g.capture_begin(pool=self.segments[0].pool(), capture_error_mode="thread_local")

        import gc
        gc.collect()
        torch.cuda.synchronize()
        dp.drain_watchdog()       # nothing of the warm-up's collectives may be left with ProcessGroupNCCL's watchdog thread when the capture begins (dp.py)
        if getattr(self, "_capture_stream", None) is None:      # one stream for every capture of this loop (autograd's accumulation nodes remember the stream they were made on)
            self._capture_stream = torch.cuda.Stream()
        capture_stream = self._capture_stream
        capture_stream.wait_stream(torch.cuda.current_stream())
        prev = dp.set_boundary(boundary)
        try:
            with torch.cuda.stream(capture_stream):
                self.segments[0].capture_begin(capture_error_mode="thread_local")
                self.out = self._forward_backward()
                self.exchange(self.sink.G)
                self._optimise_and_march()
                if open_capture[0]:
                    self.segments[-1].capture_end()
        finally:
            dp.set_boundary(prev)
        torch.cuda.current_stream().wait_stream(capture_stream)
        fo.forget_plan_events()              # (events recorded in the finished capture: not to be waited on by whatever comes next)
        self.model._presum_event = None
        capacity_rows = [(model.local_step - 2) % 16, (model.local_step - 1) % 16]
        self.content_capacity = cap_content
        if self.marched is not None:        # only the content render used the ring during the capture
            capacity_rows = [(model.local_step - 1) % 16]
        # the counters are written in issue order: with a side stream train_step issues the content render first
        return {"segments": self.segments, "between": self.between, "graphs": tuple(self.segments), "out": self.out, "capacity_rows": capacity_rows,
                "capacities": [cap_block, cap_content] if self._blocks_issued_first() else [cap_content, cap_block]}

    def step(self, message, data=None, next_data=None, next_message=None):
        """message: CPU float tensor of 0./1.; data: optional new rays/images of THIS step, next_data: of the next one (same
        shapes; see march_ahead).  Returns the static output tuple of train_step (valid until the next step; values are ready
        when the stream reaches them)."""
        if self.graphs is None:
            if data is not None:
                self._set_inputs(message, data)
                data = None
            self.prepare(message)
        unannounced = False
        if self.presum_in_adam:
            msg_cpu = message.detach().to("cpu", torch.float32)
            unannounced = self._s_for is None or not torch.equal(self._s_for, msg_cpu)
        self._set_inputs(message, data, next_data, next_message, eager_copy=unannounced or not self.stage_in_graph)
        if self.presum_in_adam:
            if unannounced:
                self.model.prepare_message(self.msg_dev)     # not announced one step early: the stand-alone pass, before the replay
            self._s_for = None if next_message is None else next_message.detach().to("cpu", torch.float32).clone()
        if self.distortion is not None and self.distortion.name == "scaling":      # this step's factor (host-side draw) and the capture of its width
            self.distortion.set_scaling(host_uniform(self.distortion.seed, self._replays, 0.75, 1.25))
            self._select(self.distortion.out_width(self.data["watermark"]["rays_o_block"].shape[2]))
        for i, g in enumerate(self.segments):
            g.replay()
            if i == 0:
                slot = self._replays % len(self.msg_ring)
                self._replays += 1       # immediately: the device counter has advanced whatever happens to the rest of this step
                if self.stage_in_graph:  # the ring row of this step may be rewritten once this replay's opening kernel has read it
                    ev = torch.cuda.Event()
                    ev.record()
                    self.msg_events[slot] = ev
            if i < len(self.between):
                self.between[i]()        # the collective between two segments (RCCL; ordered after the segment on this stream)
        self.steps_done += 1
        if self.opt_shard is not None:
            self.model._codebook_stale = True      # tables of the other ranks' bits: stale here until gather_codebook()
        return self.out

    def ensure_capacity(self, growth=1.25):
        """One host read: if the last replay produced more points than the buffers hold (its overflowing rays were dropped, like the
        reference's bounded mode), re-size from the current rays with `growth` times the headroom and capture again.  Returns True when
        it re-captured.  Callers that draw new rays every step call this now and then, outside their timed region."""
        if self.graphs is None:
            return False
        over = bool(self.overflowed())
        if exchange_active():       # re-capturing runs warm-up steps with collectives in them: every rank or none
            import torch.distributed as dist
            flag = torch.tensor([1.0 if over else 0.0], device=self.device if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            over = bool(flag.item() > 0)
        if not over:
            return False
        self.headroom = (1.0 + self.headroom) * growth - 1.0
        self.content_headroom = (1.0 + self.content_headroom) * growth - 1.0
        message = self.msg_dev.detach().to("cpu", torch.float32)
        torch.cuda.synchronize()
        self.graphs, self.segments, self.between, self.marched = None, [], [], None
        self.model.drop_marched()
        self._s_for = None
        self.prepare(message)
        return True

    def overflowed(self):
        """True if the last replay produced more points than the buffers hold (one host read of two counters)."""
        if self.marched is not None and len(self.marched) > 1:
            return any(int(r["counter"][0]) > r["capacity"] for r in self.marched)
        if self.marched is not None:
            return int(self.marched[0]["counter"][0]) > self.marched[0]["capacity"] or \
                int(self.model.step_counter[self.capacity_rows[-1], 0]) > self.content_capacity
        totals = self.model.step_counter[self.capacity_rows, 0].tolist()
        return any(t > c for t, c in zip(totals, self.capacities))
