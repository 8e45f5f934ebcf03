"""CodebookAdam: torch.optim.Adam (the optimiser the reference builds, main_nerf_wtmk.py:110) whose update of the
selected codebook tables is one native pass (opt_codebook_adam).

Every selected table carries the same gradient G (csrc/hashgrid.hip), so instead of materialising D dense
gradients and running the generic multi-tensor Adam over them (~10 passes over 128 MiB), one kernel reads G once
and updates param / exp_avg / exp_avg_sq of the D tables in place.  Semantics are torch.optim.Adam's: per-table
step counts (a table's step advances only when it is selected, exactly like a parameter whose grad is None is
skipped), bias correction, eps outside the square root, no weight decay / amsgrad.  State lives in `self.state`
in torch's own format, so `state_dict()` / `load_state_dict()` interoperate with a plain Adam checkpoint.
Parameters that have an ordinary `.grad` (the decoder) are handled by the inherited `step()`."""
import ctypes
import math

import torch

from . import _native as nv


def fused_shared_step(optimizer, group, selected, G, grad_scale=1.0):
    """torch.optim.Adam's update of the `selected` tables of `group`, all carrying the gradient G [T,2], as ONE pass (opt_codebook_adam): state in
    `optimizer.state` in torch's own (non-capturable) format, per-table step counts."""
    beta1, beta2 = group["betas"]
    lr, eps = float(group["lr"]), float(group["eps"])
    D = len(selected)
    vp = ctypes.c_void_p * D
    pp, pm, pv, step_sizes, inv_bc2 = vp(), vp(), vp(), (ctypes.c_float * D)(), (ctypes.c_float * D)()
    handles = optimizer.__dict__.setdefault("_nsig_table_handles", {})      # id(table) -> (table, state dict, exp_avg, exp_avg_sq, their addresses, step count)
    state = optimizer.state
    for i, p in enumerate(selected):
        h = handles.get(id(p))
        # validated against the LIVE state: optimizer.load_state_dict() replaces the inner dicts and their tensors (a cached dict would
        # keep pointing at the orphaned moments)
        if h is None or h[0] is not p or state.get(p) is not h[1] or h[1].get("exp_avg") is not h[2] or h[1].get("exp_avg_sq") is not h[3] or h[1].get("step") is not h[8]:
            st = optimizer.state[p]
            if len(st) == 0:   # torch.optim.Adam._init_group
                st["step"] = torch.tensor(0.0, dtype=torch.float32)
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            for t in (p, st["exp_avg"], st["exp_avg_sq"]):
                if not (t.is_contiguous() and t.dtype == torch.float32):
                    raise ValueError("fused_shared_step: tables and their Adam moments must be contiguous float32 tensors")
            h = handles[id(p)] = (p, st, st["exp_avg"], st["exp_avg_sq"], p.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(),
                                  None if st["step"].is_cuda else st["step"].numpy(), st["step"])
        pp[i], pm[i], pv[i] = h[4], h[5], h[6]
        count = h[7]                     # the step count, torch's host tensor seen through numpy: += 1 without a dispatcher call (D of them per step)
        if count is None:                # (a count left on the device by the captured loop: one synchronising read each)
            count = h[8]
        count += 1
        step = float(count)
        step_sizes[i] = lr / (1.0 - beta1 ** step)
        inv_bc2[i] = 1.0 / math.sqrt(1.0 - beta2 ** step)
    nv.call("opt_codebook_adam", nv.ptr(G), pp, pm, pv, D, float(beta1), float(beta2), eps, step_sizes, inv_bc2, float(grad_scale), nv.stream())
    _bump_versions(selected)


_SHARED_SINKS = None      # weak set of fieldops.SharedGradient sinks with (possibly) pending gradients


def install_shared_gradient_hook(sink):
    """Register `sink` (a fieldops.SharedGradient) with the process-wide optimiser pre-step hook (installed on first use): whenever ANY optimiser is about to
    step, a sink whose carrier table it manages is served -- by the fused pass where the optimiser is a plain Adam (the reference's,
    main_nerf_wtmk.py:110: no weight decay / amsgrad / maximize, not capturable), by dissolving it into ordinary per-table gradients otherwise."""
    global _SHARED_SINKS
    import weakref
    if _SHARED_SINKS is None:
        _SHARED_SINKS = weakref.WeakSet()
        from torch.optim.optimizer import register_optimizer_step_post_hook, register_optimizer_step_pre_hook
        register_optimizer_step_pre_hook(_shared_gradient_pre_step)
        register_optimizer_step_post_hook(_dense_takeover_post_step)
    _SHARED_SINKS.add(sink)


@torch.no_grad()
def _shared_gradient_pre_step(optimizer, args, kwargs):
    for sink in list(_SHARED_SINKS or ()):
        if not sink.pending():
            continue
        members = optimizer.__dict__.setdefault("_nsig_group_members", {})      # per param group: the ids of its parameters (rebuilt when the group's size changes)
        group = None
        for g in optimizer.param_groups:
            ids = members.get(id(g))
            if ids is None or ids[0] != len(g["params"]):
                ids = members[id(g)] = (len(g["params"]), {id(q) for q in g["params"]})
            if id(sink.carrier) in ids[1]:
                group = g
                break
        if group is None:
            continue                                    # not this optimiser's parameter
        plain = (isinstance(optimizer, torch.optim.Adam) and not group.get("weight_decay", 0) and not group.get("amsgrad", False) and not group.get("maximize", False)
                 and not group.get("capturable", False) and not group.get("differentiable", False) and not torch.is_tensor(group["lr"])
                 and all(id(t) in ids[1] for t in sink.live))
        if not plain:
            sink.dissolve()                             # the optimiser sees D ordinary dense gradients
            continue
        fused_shared_step(optimizer, group, sink.live, sink.G)
        sink.consumed()                                 # `.grad` of every selected table is None now: the optimiser's own loop skips them
        if sink.dense_takeover:
            _dense_takeover(optimizer)


def _dense_takeover(optimizer):
    """The rest of the step of a plain torch.optim.Adam -- every parameter that carries an ordinary dense CUDA float32 gradient: the decoder's 27 tensors --
    as ONE launch (opt_adam_dense_host) instead of the optimiser's multi-tensor launches and their Python: same arithmetic, state in torch's own
    non-capturable format (host step counts), per group hyper-parameters.  The gradients are set aside for the duration of optimizer.step() -- the optimiser's
    own loop, which runs right behind this hook, then finds nothing left to do -- and put back by the post-step hook: `p.grad` reads the same after the step.
    Groups or tensors this does not cover (weight decay, amsgrad, capturable, CPU / non-fp32 / non-contiguous tensors) are left to the optimiser."""
    aside = []
    for group in optimizer.param_groups:
        if (group.get("weight_decay", 0) or group.get("amsgrad", False) or group.get("maximize", False) or group.get("capturable", False)
                or group.get("differentiable", False) or torch.is_tensor(group["lr"])):
            continue
        ps = [p for p in group["params"] if p.grad is not None and p.is_cuda and p.dtype == torch.float32 and p.grad.dtype == torch.float32
              and not p.grad.is_sparse and p.is_contiguous() and p.grad.is_contiguous()]
        if not ps:
            continue
        beta1, beta2 = group["betas"]
        lr, eps = float(group["lr"]), float(group["eps"])
        handles = optimizer.__dict__.setdefault("_nsig_dense_handles", {})      # id(p) -> (p, state dict, exp_avg, exp_avg_sq, step count as numpy, step tensor)
        n = len(ps)
        vp, fl = ctypes.c_void_p * n, ctypes.c_float * n
        pp, pg, pm, pv, numel, ss, ib = vp(), vp(), vp(), vp(), (ctypes.c_uint32 * n)(), fl(), fl()
        skip = False
        for i, p in enumerate(ps):
            h = handles.get(id(p))
            if (h is None or h[0] is not p or optimizer.state.get(p) is not h[1] or h[1].get("exp_avg") is not h[2] or h[1].get("exp_avg_sq") is not h[3]
                    or h[1].get("step") is not h[5]):      # (live state, see fused_shared_step)
                st = optimizer.state[p]
                if len(st) == 0:   # torch.optim.Adam._init_group
                    st["step"] = torch.tensor(0.0, dtype=torch.float32)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                if st["step"].is_cuda or not (st["exp_avg"].is_contiguous() and st["exp_avg_sq"].is_contiguous()):
                    skip = True    # (a state format from elsewhere: the optimiser's own business)
                    break
                h = handles[id(p)] = (p, st, st["exp_avg"], st["exp_avg_sq"], st["step"].numpy(), st["step"], p.data_ptr(), st["exp_avg"].data_ptr(),
                                      st["exp_avg_sq"].data_ptr(), p.numel())
            pp[i], pm[i], pv[i], numel[i], pg[i] = h[6], h[7], h[8], h[9], p.grad.data_ptr()
        if skip:
            continue
        for i, p in enumerate(ps):
            count = handles[id(p)][4]
            count += 1
            k = float(count)
            ss[i] = lr / (1.0 - beta1 ** k)
            ib[i] = 1.0 / math.sqrt(1.0 - beta2 ** k)
        nv.call("opt_adam_dense_host", n, pp, pg, pm, pv, numel, ss, ib, float(beta1), float(beta2), eps, 1.0, nv.stream())
        _bump_versions(ps)
        for p in ps:
            aside.append((p, p.grad))
            p.grad = None
    if aside:
        optimizer.__dict__["_nsig_grads_aside"] = aside


def _dense_takeover_post_step(optimizer, args, kwargs):
    aside = optimizer.__dict__.pop("_nsig_grads_aside", None)
    if aside:
        for p, g in aside:
            if p.grad is None:
                p.grad = g


class CodebookAdam(torch.optim.Adam):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, **kw):
        if kw.get("weight_decay", 0) or kw.get("amsgrad", False) or kw.get("maximize", False):
            raise NotImplementedError("CodebookAdam implements plain Adam (the reference's configuration)")
        super().__init__(params, lr=lr, betas=betas, eps=eps, **kw)

    def _group_of(self, p):
        for g in self.param_groups:
            if any(q is p for q in g["params"]):
                return g
        raise ValueError("parameter is not managed by this optimiser")

    @torch.no_grad()
    def step_shared(self, selected, G, grad_scale=1.0):
        """Adam step of the `selected` tables (all in one param group) with the shared gradient G [T,2]."""
        if not selected:
            return
        fused_shared_step(self, self._group_of(selected[0]), selected, G, grad_scale)


def _prepare_device_state(opt, tables):
    """Allocate Adam state for every table up front (static addresses for graph replay); torch's capturable format."""
    for t in tables:
        st = opt.state[t]
        if len(st) == 0:
            st["step"] = torch.zeros((), dtype=torch.float32, device=t.device)
            st["exp_avg"] = torch.zeros_like(t, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(t, memory_format=torch.preserve_format)
        elif not st["step"].is_cuda:
            st["step"] = st["step"].to(t.device)


def _step_shared_sel(opt, tables, message_dev, G, lr_dev, grad_scale=1.0, next_message_dev=None, S_next=None):
    """Adam step of table 2i + message[i] for every bit, everything message-dependent resolved on the device.
    next_message_dev + S_next: the same pass also writes the pre-summed codebook of the next step's message into S_next
    (opt_codebook_adam_sel_next)."""
    group = opt._group_of(tables[0])
    beta1, beta2 = group["betas"]
    cache = getattr(opt, "_sel_cache", None)
    if (cache is None or len(cache[0]) != len(tables) or any(a is not b for a, b in zip(cache[0], tables))
            or any(opt.state.get(t) is not st for t, st in zip(tables, cache[4]))):      # (load_state_dict replaces the state: new addresses)
        _prepare_device_state(opt, tables)
        D = len(tables) // 2
        arrays = (nv.ptr_array([t.data for t in tables]), nv.ptr_array([opt.state[t]["exp_avg"] for t in tables]),
                  nv.ptr_array([opt.state[t]["exp_avg_sq"] for t in tables]), nv.ptr_array([opt.state[t]["step"] for t in tables]))
        scratch = torch.empty(2 * D, dtype=torch.float32, device=tables[0].device)
        cache = opt._sel_cache = (list(tables), arrays, scratch, D, [opt.state[t] for t in tables])
    _, (pp, pm, pv, ps), scratch, D, _ = cache
    if next_message_dev is not None:
        if S_next is None or S_next.dtype != torch.float32 or not S_next.is_contiguous() or S_next.numel() != tables[0].numel():
            raise ValueError("step_shared_sel: S_next must be a contiguous float32 tensor of one table's size")
        nv.call("opt_codebook_adam_sel_next", nv.ptr(G), pp, pm, pv, ps, nv.ptr(message_dev), D, nv.ptr(lr_dev), float(beta1), float(beta2),
                float(group["eps"]), float(grad_scale), nv.ptr(scratch), nv.ptr(next_message_dev), nv.ptr(S_next), nv.stream())
    else:
        nv.call("opt_codebook_adam_sel", nv.ptr(G), pp, pm, pv, ps, nv.ptr(message_dev), D, nv.ptr(lr_dev), float(beta1), float(beta2),
                float(group["eps"]), float(grad_scale), nv.ptr(scratch), nv.stream())
    _bump_versions(tables)


CodebookAdam.step_shared_sel = torch.no_grad()(_step_shared_sel)


def _step_dense(opt, lr_dev, grad_scale=1.0):
    """torch.optim.Adam's update of every parameter that carries a dense `.grad` (the decoder) through opt_adam_dense: one
    pass of 1024-element chunks instead of the generic multi-tensor kernel's 64K-element ones.  State in torch's capturable
    format (device step counts), so `step()` and `state_dict()` keep working on it."""
    todo = []
    for group in opt.param_groups:
        for p in group["params"]:
            if p.grad is None:
                continue
            if not (p.is_cuda and p.dtype == torch.float32 and p.grad.dtype == torch.float32 and p.is_contiguous() and p.grad.is_contiguous()):
                raise NotImplementedError("step_dense handles contiguous float32 CUDA parameters")
            st = opt.state[p]
            if len(st) == 0:
                st["step"] = torch.zeros((), dtype=torch.float32, device=p.device)
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            elif not st["step"].is_cuda:
                st["step"] = st["step"].to(p.device)
            todo.append((group, p))
    if not todo:
        return
    group = todo[0][0]
    if any(g["betas"] != group["betas"] or g["eps"] != group["eps"] for g, _ in todo):
        raise NotImplementedError("step_dense expects one (betas, eps) for all dense parameters")
    ps = [p for _, p in todo]
    n = len(ps)
    scratch = getattr(opt, "_dense_scratch", None)
    if scratch is None or scratch.numel() < 64 * ((n + 31) // 32):
        scratch = opt._dense_scratch = torch.empty(64 * ((n + 31) // 32), dtype=torch.float32, device=ps[0].device)
    numel = (ctypes.c_uint32 * n)(*[p.numel() for p in ps])
    nv.call("opt_adam_dense", n, nv.ptr_array([p.data for p in ps]), nv.ptr_array([p.grad for p in ps]),
            nv.ptr_array([opt.state[p]["exp_avg"] for p in ps]), nv.ptr_array([opt.state[p]["exp_avg_sq"] for p in ps]),
            nv.ptr_array([opt.state[p]["step"] for p in ps]), numel, nv.ptr(lr_dev), float(group["betas"][0]), float(group["betas"][1]),
            float(group["eps"]), float(grad_scale), nv.ptr(scratch), nv.stream())
    _bump_versions(ps)


CodebookAdam.step_dense = torch.no_grad()(_step_dense)


def _bump_versions(tensors):
    """The native in-place update is invisible to torch's version counters; caches keyed on them (the pre-summed
    codebook in NeRFNetwork) must see the change."""
    setter = getattr(torch._C._autograd, "_unsafe_set_version_counter", None)
    if setter is not None:
        setter(list(tensors), [t._version + 1 for t in tensors])
    else:  # pragma: no cover - older torch: an in-place no-op bumps the counter
        for t in tensors:
            t.add_(0)
