"""Checkpoint compatibility: the on-disk format of the reference's Trainer
(/root/reference/nerf/utils_wtmk_disen.py:1385-1517) -- a torch.save'd dict with 'epoch', 'global_step', 'stats',
'mean_count', 'mean_density', optionally 'optimizer' / 'lr_scheduler' / 'scaler', and 'model' = state_dict().

The model keeps the reference's parameter surface (names, shapes, fp32), so no re-layout happens here; what this module
adds is the dict around it and the tolerant loading rules: a bare state_dict is accepted (:1469-1472), loading is
strict=False with the missing / unexpected keys reported (:1474-1479) -- a clean stage-1 checkpoint has no `msg_encoder.*`
/ `msg_decoder.*` keys -- tinycudann parameters stored in half precision are widened to fp32, and the derived device
images of the model (packed MLP weights, pre-summed codebook) are invalidated."""
import os

import torch


def gather_sharded_codebook(model, gather=True):
    """With the codebook optimiser sharded over the ranks (dp.optimizer_shard; on by default from four ranks) a rank's copies of the tables
    and Adam moments it does not own are stale after every captured step until GraphedWatermarkLoop.gather_codebook() has run.  Whoever
    reads the tables on the host side -- a checkpoint, an evaluation through the host-side selection -- calls this first.  gather=True
    brings them up to date (a COLLECTIVE: every rank must call it, also the ranks that do not write the file); gather=False refuses."""
    if not getattr(model, "_codebook_stale", False):
        return
    loops = [l for l in getattr(model, "_graphed_loops", ()) if getattr(l, "opt_shard", None) is not None]
    if not gather or not loops:
        raise RuntimeError("the codebook optimiser is sharded over the ranks and this rank's copies of the other ranks' tables are stale: "
                           "call GraphedWatermarkLoop.gather_codebook() on EVERY rank first (checkpoint_state(..., gather=True) does)")
    for loop in loops:
        loop.gather_codebook()


def checkpoint_state(model, epoch=0, global_step=0, stats=None, optimizer=None, lr_scheduler=None, scaler=None, full=False, gather=True):
    """gather: see gather_sharded_codebook -- with a sharded codebook optimiser this is a collective call (all ranks enter it, rank 0 saves)."""
    gather_sharded_codebook(model, gather)
    state = {"epoch": epoch, "global_step": global_step, "stats": stats if stats is not None else {}}
    if getattr(model, "cuda_ray", False):
        state["mean_count"] = model.mean_count
        state["mean_density"] = model.mean_density
    if full:
        if optimizer is not None:
            state["optimizer"] = optimizer.state_dict()
        if lr_scheduler is not None:
            state["lr_scheduler"] = lr_scheduler.state_dict()
        if scaler is not None:
            state["scaler"] = scaler.state_dict()
    state["model"] = model.state_dict()
    return state


def save_checkpoint(path, model, **kw):
    """Writes `<path>` (e.g. <workspace>/checkpoints/ngp_ep0010.pth) in the reference's format.  More than one rank: rank 0 writes.

    With the codebook optimiser sharded over the ranks (on by default from four ranks) the tables are gathered first -- a COLLECTIVE, so
    every rank has to make this call, not only rank 0 as in the reference's `if self.local_rank == 0: self.save_checkpoint(...)`; the
    other ranks take part in the gather, skip the write and leave through a barrier, i.e. when the file exists.  A caller that cannot
    arrange that passes gather=False: the call then raises on stale tables instead of entering a collective alone (and blocking until the
    process group's timeout)."""
    import torch.distributed as dist
    multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    gathered = multi and bool(getattr(model, "_codebook_stale", False)) and kw.get("gather", True)
    state = checkpoint_state(model, **kw)            # (gathers the sharded tables: all ranks)
    if not multi or dist.get_rank() == 0:
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        torch.save(state, path)
    if gathered:                                     # every rank came through the gather above, so every rank reaches this barrier
        dist.barrier()
    return path


def load_checkpoint(path_or_dict, model, optimizer=None, lr_scheduler=None, scaler=None, model_only=False, map_location=None):
    """Returns (missing_keys, unexpected_keys, meta) where meta holds epoch / global_step / stats when present."""
    ckpt = torch.load(path_or_dict, map_location=map_location, weights_only=False) if isinstance(path_or_dict, (str, os.PathLike)) else path_or_dict
    sd = ckpt["model"] if "model" in ckpt else ckpt
    own = model.state_dict()
    fixed = {}
    for k, v in sd.items():
        if k in own and torch.is_tensor(v) and v.is_floating_point() and v.dtype != own[k].dtype:
            v = v.to(own[k].dtype)          # e.g. fp16 tcnn params
        fixed[k] = v
    missing, unexpected = model.load_state_dict(fixed, strict=False)
    invalidate_derived(model)
    meta = {}
    if "model" in ckpt:
        if getattr(model, "cuda_ray", False):
            if "mean_count" in ckpt:
                model.mean_count = ckpt["mean_count"]
            if "mean_density" in ckpt:
                model.mean_density = ckpt["mean_density"]
        if not model_only:
            meta = {k: ckpt[k] for k in ("epoch", "global_step", "stats") if k in ckpt}
            # like the reference (:1497-1517): a section that does not fit (e.g. a plain-Adam state of another parameter grouping)
            # is reported and skipped, the model-only resume continues
            for obj, key in ((optimizer, "optimizer"), (lr_scheduler, "lr_scheduler"), (scaler, "scaler")):
                if obj is not None and key in ckpt:
                    try:
                        if key == "optimizer":
                            _load_optimizer(obj, ckpt[key])
                        else:
                            obj.load_state_dict(ckpt[key])
                    except Exception as e:          # noqa: BLE001 -- the reference catches everything here and warns
                        meta.setdefault("skipped", []).append(key)
                        print(f"[WARN] Failed to load {key}: {type(e).__name__}: {e}")
    return list(missing), list(unexpected), meta


def invalidate_derived(model):
    """The model's derived device images after its parameters were overwritten.  Both BUFFERS -- the packed MLP weights and the
    pre-summed codebook -- are kept and only their keys invalidated: a captured GraphedWatermarkLoop holds them by raw address
    (field_fwd's weight image; adopt_presum, S_next of opt_codebook_adam_sel_next), so returning them to the allocator would leave
    later replays reading and writing freed memory.  A loop registered on the model re-packs the weights now (no replay would) and
    forgets which message the pre-sum belongs to."""
    for attr in ("_packed_cache", "_presum_cache"):
        cache = getattr(model, attr, None)
        if cache is not None:
            setattr(model, attr, (None, cache[1]))
    for loop in list(getattr(model, "_graphed_loops", ())):
        loop.invalidate()


def _load_optimizer(optimizer, state):
    """optimizer.load_state_dict that keeps every existing state tensor and device scalar IN PLACE.  A captured
    GraphedWatermarkLoop holds the addresses of exp_avg / exp_avg_sq / step and of the learning-rate tensor it installed into the
    param groups; torch's load_state_dict would replace them by fresh tensors (and the tensor lr by a float), leaving the replays
    updating freed memory and detached from the loaded values.  Here the loaded values are copied into the old storage; state the
    checkpoint does not hold for a parameter that had some is reset to zero (Adam's initial state) in the old storage."""
    lr_tensors = [g["lr"] if torch.is_tensor(g.get("lr")) else None for g in optimizer.param_groups]
    old = {p: dict(st) for p, st in optimizer.state.items() if len(st)}
    optimizer.load_state_dict(state)
    for g, lr_dev in zip(optimizer.param_groups, lr_tensors):
        if lr_dev is not None:
            lr_dev.fill_(float(g["lr"]))
            g["lr"] = lr_dev
    for p, before in old.items():
        st = optimizer.state[p]
        for k, ov in before.items():
            if not torch.is_tensor(ov):
                continue
            nv_ = st.get(k)
            if torch.is_tensor(nv_) and nv_.shape == ov.shape:
                ov.copy_(nv_.to(ov.device, ov.dtype))
            elif nv_ is None:
                ov.zero_()
            else:
                continue
            st[k] = ov
