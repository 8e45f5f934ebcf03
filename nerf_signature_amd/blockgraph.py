"""The block render + decoder of an EAGER training step -- forward and backward -- as two hipGraphs.

Under somebody else's training loop (the reference Trainer's, utils_wtmk_disen.py:1164-1190, on top of the drop-in modules) a step is host-bound:
~100 launches issued one by one from Python.  The half of it that belongs to the watermark blocks has static shapes: the block rays are one pair of
tensors per dataset (provider_wtmk.py:442-494; kept samples + base-level planes after their second sighting, NeRFNetwork.fix_rays), the decoder's
input is [D, bh, bw, 3] every step, and the only thing a step changes in it is the CONTENT of buffers with fixed addresses -- the pre-summed
codebook of the step's message (computed eagerly in front of the replay), the codebook level's plane, the decoder's parameters.  So those
~30 launches (codebook plane, both MLPs, compositing, the decoder chain; and their backward: decoder chain, compositing, MLP backward, scatter into
the shared gradient, the decoder's parameter gradients) are captured once and replayed as two graph launches, in the manner of
torch.cuda.make_graphed_callables: the forward capture builds the autograd graph, the backward capture runs torch.autograd.grad over it.
The content render (a new pose and new pixels every step: its sample count varies) stays eager.

Same kernels, same arguments, same order on the same stream as the eager calls they replace: results are bit-identical
(tests/test_gpu_amp_ckpt.py::test_block_graph_*).  Anything the capture does not cover makes run() return None and the caller takes the eager route:
the `scaling` distortion (a new decoder input width every step), a model without the shared-gradient sink, rays that are not (yet) kept, more than one
rank, an outer capture.
NERFSIG_DROPIN_OFF=block_graph turns it off."""
import os

import torch

from . import fieldops as fo
from .switches import dropin_off
from . import hidden_models
from .hidden_models import normalize_img


class _Replay(torch.autograd.Function):
    """One autograd node for both captured halves.  Inputs: one selected table (so that the node is recorded even with a frozen decoder; its
    gradient travels through the shared-gradient sink, not through autograd) and the decoder's parameters."""

    @staticmethod
    def forward(ctx, graph, selected, anchor, *params):
        graph.forward_graph.replay()
        graph.generation += 1
        ctx.graph, ctx.generation, ctx.selected, ctx.n = graph, graph.generation, selected, len(params)
        decoded, pred = graph.decoded.clone(), graph.pred.clone()      # (static buffers: the next replay overwrites them)
        ctx.mark_non_differentiable(pred)
        ctx.set_materialize_grads(False)
        return decoded, pred

    @staticmethod
    def backward(ctx, grad_decoded, _=None):
        g = ctx.graph
        if grad_decoded is None:
            return (None,) * (3 + ctx.n)
        if ctx.generation != g.generation:
            raise RuntimeError("BlockDecodeGraph: backward of a forward pass whose saved activations a later forward has overwritten (the captured block render "
                               "keeps ONE set of activations: run each step's backward before the next step's forward, or set NERFSIG_DROPIN_OFF=block_graph)")
        if g.backward_of == ctx.generation:
            raise RuntimeError("BlockDecodeGraph: second backward through the same captured forward pass (retain_graph is not supported: NERFSIG_DROPIN_OFF=block_graph)")
        g.backward_of = ctx.generation
        g.sink.begin_accumulation(ctx.selected)           # (Python state of the shared gradient: what _FieldFunction.backward does in front of its scatter)
        g.seed.copy_(grad_decoded.reshape(g.seed.shape))
        g.backward_graph.replay()
        flat = g.flat.clone()
        grads, off = [], 0
        for p, wanted in zip(g.params, ctx.needs_input_grad[3:]):
            n = p.numel()
            grads.append(flat[off:off + n].view_as(p) if wanted else None)
            off += n
        return (None, None, None, *grads)


class BlockDecodeGraph:
    def __init__(self, eager_steps=3):
        self.eager_steps = int(eager_steps)      # steps seen with the same key before capturing (kept samples exist from the second sighting on)
        self.key, self.seen = None, 0
        self.forward_graph = self.backward_graph = None
        self.failed = None
        self.generation, self.backward_of = 0, -1
        self.captures = 0

    def _content_key(self, content):
        raise NotImplementedError("a BlockDecodeGraph captures the block render and the decoder only (StepGraph takes the content render too)")

    # ------------------------------------------------------------------ what must hold for the captured launches to be the eager ones

    def _record(self, model, rays_o, rays_d):
        marched = getattr(model, "_marched", None)
        if not marched:
            return None
        _, o, d = model._flatten_rays(rays_o, rays_d)
        rec = marched.get(model._rays_key(o, d))
        if rec is None or rec.get("fixed") is None or rec["grid_key"] != model.grid_key():
            return None
        if rec["fixed"].key != fo.FixedPoints._tables_key(model.encoder.tables()):
            return None                                     # a base table moved: the eager route refreshes the kept planes in place
        return rec

    def _key(self, model, rays_o, rays_d, rec, fused, kw, sink):
        sp, cp = model.sigma_net.params, model.color_net.params
        return (id(model), id(rays_o), id(rays_d), rays_o._version, rays_d._version, tuple(rays_o.shape), id(rec), id(rec["fixed"]), rec["xyzs"].data_ptr(),
                rec["capacity"], rec["fixed"].planes.data_ptr(), tuple(p.data_ptr() for p in fused[1]), tuple(tuple(p.shape) for p in fused[1]),
                tuple(bool(p.requires_grad) for p in fused[1]), float(fused[0]), sp.data_ptr(), sp._version, cp.data_ptr(), cp._version,
                model._packed().data_ptr(), id(sink), sink.G.data_ptr(),
                None if model._presum_cache is None else model._presum_cache[1].data_ptr(),
                float(kw.get("dt_gamma", 0)), int(kw.get("max_steps", 1024)), float(kw.get("T_thresh", 1e-4)), int(fo.nv.fn("mlp_get_precision")()),
                float(model.density_scale), float(model.bound))

    def run(self, model, rays_o, rays_d, message, render_kwargs, distortion=None, content=None):
        """(decoded [D, 1], clamped blocks [D, bh, bw, 3]) of this step through the captured launches, or None: take the eager route.
        content (StepGraph only): (content rays_o, rays_d, images, lambda_w, lambda_i) -- the whole forward pass is then captured and the result is
        train_step's six return values.
        distortion: None or a distortion.DistortionLayer whose owner has drawn this step's parameters into its (static) device buffers; every kind but
        `scaling` (which changes the decoder's input width from step to step) is part of the captured forward."""
        if self.failed is not None or dropin_off("block_graph"):
            return None
        if distortion is not None and (distortion.name == "scaling" or distortion.param is None or not distortion.param.is_cuda):
            return None
        if not (torch.is_tensor(message) and rays_o.is_cuda and rays_o.dim() == 4 and model.training and torch.is_grad_enabled() and model.cuda_ray
                and not torch.cuda.is_current_stream_capturing() and model.normalization is normalize_img and model.grad_sink is None
                and getattr(model, "shared_gradient_step", False) and not getattr(model, "device_select", False)
                and getattr(model, "point_capacity", None) is None):
            return None
        from .network import _data_parallel
        if _data_parallel():          # (more than one rank: DistributedDataParallel-style loops reduce autograd's own gradients)
            return None
        rec = self._record(model, rays_o, rays_d)
        if rec is None:
            self.seen = 0
            return None
        D, H, W = rays_o.shape[0], rays_o.shape[1], rays_o.shape[2]
        fused = model.msg_decoder._fused_params(D, 3, H, W, rays_o) if hasattr(model.msg_decoder, "_fused_params") and rays_o.dtype == torch.float32 else None
        sink = model._shared_sink
        if fused is None or sink is None or sink.G.device != rays_o.device or message.numel() != model.message_dim:
            return None
        kw = dict(render_kwargs)
        key = self._key(model, rays_o, rays_d, rec, fused, kw, sink) + (
            (None,) if distortion is None else (distortion.name, distortion.param.data_ptr(), None if distortion.noise is None else distortion.noise.data_ptr()))
        if content is not None:
            key = key + self._content_key(content)
        if key != self.key:
            self.key, self.seen = key, 0
            self.forward_graph = self.backward_graph = None
        selected, bits, _ = model._select(message)          # this step's pre-summed codebook, into the buffer the captured launches read (one read of the bits, as eager)
        anchor = next((t for t in selected if t.requires_grad), None)
        if anchor is None:
            return None
        if content is not None and self.seen >= self.eager_steps and not self._stage_content(model, content, message, kw):
            return None                                     # (more samples than the captured buffers hold: this step runs eagerly, the next capture is larger)
        if self.forward_graph is None:
            self.seen += 1
            if self.seen <= self.eager_steps:
                return None
            try:
                self._capture(model, rays_o, rays_d, message, kw, fused, sink, selected, distortion, content)
            except Exception as e:      # noqa: BLE001 -- whatever it was: never again in this process, and say so
                self.failed = repr(e)
                self.forward_graph = self.backward_graph = None
                torch.cuda.synchronize()
                import warnings
                warnings.warn(f"BlockDecodeGraph: capture failed ({self.failed}); the eager route is used from here on")
                return None
        self.sink = sink
        self.params = list(fused[1])
        if content is not None:
            loss, lossi, lossw, pred, cpred = _ReplayStep.apply(self, list(selected), anchor, *self.params)
            return pred, content[2], cpred, lossi, lossw, loss
        return _Replay.apply(self, list(selected), anchor, *self.params)

    # ------------------------------------------------------------------ capture

    def _capture(self, model, rays_o, rays_d, message, kw, fused, sink, selected, distortion=None, content=None):
        kw = dict(kw)
        kw.update(staged=False, bg_color=1, perturb=False, force_all_rays=True)
        params = list(fused[1])
        n_flat = sum(p.numel() for p in params)
        dev = rays_o.device
        self.params = params
        self.flat = torch.empty(n_flat, dtype=torch.float32, device=dev)           # the decoder's parameter gradients of the last backward replay
        # the backward's seed, copied in before its replay: d loss / d decoded -- or, with the whole step captured, the (scalar) gradient of the loss itself
        self.seed = torch.empty((rays_o.shape[0], 1) if content is None else (), dtype=torch.float32, device=dev)
        torch.cuda.synchronize()
        model._presum_event = None                 # (recorded outside the capture: the captured render must not wait on it)
        fo.forget_plan_events()
        pending_before = sink.pending()
        self.forward_graph, self.backward_graph = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        if getattr(self, "_stream", None) is None:
            self._stream = torch.cuda.Stream()
        stream = self._stream
        stream.wait_stream(torch.cuda.current_stream())
        anchor = next(t for t in selected if t.requires_grad)
        with torch.enable_grad(), torch.cuda.stream(stream):
            # The decoder's parameters enter the captured passes through ALIASES (same storage, new autograd leaves made on the capture stream): the
            # gradient accumulators of the real Parameters were created by the eager steps on the caller's stream -- usually the default stream -- and
            # autograd synchronises a leaf's accumulator stream with the producer of its gradient: that would pull the default stream into the capture.
            aliases = [p.detach().requires_grad_(p.requires_grad) for p in params]
            self.forward_graph.capture_begin(capture_error_mode="thread_local")
            try:
                image = model.render(rays_o, rays_d, message, **kw)["image"]
                if distortion is not None and distortion.geometric:      # (hidden_models.decode_rendered's two routes, with the aliases for parameters)
                    from .distortion import _DistortGeometry
                    resampled, pred = _DistortGeometry.apply(image, distortion.kind, distortion.param, image.shape[2])
                    decoded = hidden_models._FusedDecoder.apply(resampled, fused[0], True, None, None, *aliases)[0]
                else:
                    decoded, pred = hidden_models._FusedDecoder.apply(image, fused[0], True, distortion, None, *aliases)
                root = decoded
                if content is not None:      # the content render from the samples staged in front of the replay, and the three losses
                    from .trainer import _WatermarkLoss
                    cpred = model.render(self.c_o, self.c_d, message, **kw)["image"]
                    lossi, lossw, root = _WatermarkLoss.apply(cpred, self.c_gt, decoded, self.msg_s.unsqueeze(-1), float(content[3]), float(content[4]), 10.0)
            finally:
                self.forward_graph.capture_end()
            if not pending_before:
                sink.begin_accumulation(list(selected))      # so that the captured _FieldFunction.backward finds the accumulation open and records no zero-fill
            hidden_models.set_grad_arena(self.flat)
            self.backward_graph.capture_begin(pool=self.forward_graph.pool(), capture_error_mode="thread_local")
            try:
                torch.autograd.grad([root], [anchor] + [a for a in aliases if a.requires_grad], [self.seed], allow_unused=True)
            finally:
                self.backward_graph.capture_end()
                hidden_models.set_grad_arena(None)
                if not pending_before:
                    sink.consumed()
        torch.cuda.current_stream().wait_stream(stream)
        fo.forget_plan_events()
        model._presum_event = None
        self.decoded, self.pred = decoded.detach(), pred.detach()
        if content is not None:
            self.losses, self.cpred = (root.detach(), lossi.detach(), lossw.detach()), cpred.detach()
        self.captures += 1


class _ReplayStep(torch.autograd.Function):
    """_Replay for a StepGraph: the outputs are train_step's -- (loss, lossi, lossw, clamped blocks, content image) -- and the backward is seeded by the
    gradient of `loss` (what `scaler.scale(loss).backward()` hands down)."""

    @staticmethod
    def forward(ctx, graph, selected, anchor, *params):
        graph.forward_graph.replay()
        graph.generation += 1
        ctx.graph, ctx.generation, ctx.selected, ctx.n = graph, graph.generation, selected, len(params)
        loss, lossi, lossw = (t.clone() for t in graph.losses)
        pred, cpred = graph.pred.clone(), graph.cpred.clone()
        ctx.mark_non_differentiable(pred, cpred)
        ctx.set_materialize_grads(False)
        return loss, lossi, lossw, pred, cpred

    @staticmethod
    def backward(ctx, g_loss, g_lossi=None, g_lossw=None, *_):
        if g_lossi is not None or g_lossw is not None:
            raise NotImplementedError("StepGraph: the captured backward starts from `loss` (utils_wtmk_disen.py:1174); a backward through lossi / lossw alone needs "
                                      "NERFSIG_DROPIN_OFF=step_graph")
        return _Replay.backward(ctx, g_loss)


class StepGraph(BlockDecodeGraph):
    """BlockDecodeGraph with the rest of the step's forward and backward inside the two graphs: the content render and the losses.

    The content rays are new tensors every step and their sample count varies, so in front of every replay (a) rays, ground truth and message are copied
    into static buffers, (b) the rays are marched EAGERLY into a static sample record of fixed capacity (NeRFRenderer.march_ahead) and the count is read
    back -- the one host read an eager step has anyway.  If the samples fit, the captured forward (block render, decoder, content field pass + compositing
    over `capacity` rows, loss kernel) and later the captured backward are replayed; if they do not, THIS step runs on the eager route (nothing is dropped,
    ever) and the next capture is sized 1.3 x larger.  NERFSIG_DROPIN_OFF=step_graph: the block render + decoder alone are captured (BlockDecodeGraph)."""

    HEADROOM = 1.3

    def __init__(self, eager_steps=3):
        super().__init__(eager_steps)
        self.capacity = None
        self.overflows = 0

    def _content_key(self, content):
        o, d, gt, lw, li = content
        return (tuple(o.shape), tuple(gt.shape), float(lw), float(li), self.capacity)

    def usable_content(self, content, loss_is_bce, color_space):
        o, d, gt, _, _ = content
        return (not dropin_off("step_graph") and loss_is_bce and color_space == "srgb" and o.is_cuda and o.dtype == torch.float32
                and d.dtype == torch.float32 and gt.dtype == torch.float32 and gt.shape[-1] == 3 and tuple(gt.shape[:-1]) == tuple(o.shape[:-1]))

    def _stage_content(self, model, content, message, kw):
        from . import raymarching
        o, d, gt, _, _ = content
        if getattr(self, "c_o", None) is None or self.c_o.shape != o.shape or self.c_o.device != o.device:
            self.c_o, self.c_d, self.c_gt = torch.empty_like(o), torch.empty_like(d), torch.empty_like(gt)
            self.msg_s = torch.empty(message.numel(), dtype=torch.float32, device=o.device)
            self.capacity = None
        self.c_o.copy_(o)
        self.c_d.copy_(d)
        self.c_gt.copy_(gt)
        self.msg_s.copy_(message)
        dt_gamma, max_steps = kw.get("dt_gamma", 0), kw.get("max_steps", 1024)
        if self.capacity is None:
            self.capacity = raymarching.padded_point_count(int(model._count_points(*model._flatten_rays(self.c_o, self.c_d)[1:], dt_gamma, max_steps) * self.HEADROOM))
            self.forward_graph = self.backward_graph = None
            self.key = None              # (the capacity is part of the key: the caller's key is stale now; it is rebuilt at the next step)
        rec = model.march_ahead(self.c_o, self.c_d, dt_gamma, max_steps, capacity=self.capacity)
        n = raymarching.padded_point_count(int(rec["counter"][0]))
        if n > self.capacity or self.key is None:
            if n > self.capacity:
                self.overflows += 1
                self.capacity = raymarching.padded_point_count(int(n * self.HEADROOM))
                self.forward_graph = self.backward_graph = None
                self.key = None
            model._marched = {k: r for k, r in model._marched.items() if r is not rec}      # the eager route marches for itself
            return False
        return True

