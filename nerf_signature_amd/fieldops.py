"""Functional wrappers (tensors in, tensors out) over the hash-grid and field-network entry points of
libnerfsig (hg_*, mlp_*, field_* in include/nerfsig.h) and the autograd Functions built on them.

These replace, on the GPU, the op chains of the reference's hash_encoding.py / hash_encoding_wtmk_bit.py
and the tinycudann calls of nerf/network_wtmk_tcnn.py:97-176.  No host synchronisation happens here:
the message bits are taken from a host-side tuple (see `message_bits`), never read back from the device.
"""
import os

import torch
from torch.autograd import Function
from torch.amp import custom_bwd, custom_fwd

from . import _native as nv

T_ROWS = 1 << 19
MASK_WORDS = 6

_fwd32 = custom_fwd(device_type="cuda", cast_inputs=torch.float32)
_bwd = custom_bwd(device_type="cuda")


def message_bits(message):
    """Host tuple of 0/1 ints for a message given as a tensor (CPU: free; GPU: one small D2H copy, which is
    the single synchronisation the reference performs D times per call, hash_encoding_wtmk_bit.py:110),
    a list or a tuple."""
    if message is None:
        return None
    if isinstance(message, torch.Tensor):
        return tuple(int(v) for v in message.detach().to("cpu", torch.float32).tolist())
    return tuple(int(v) for v in message)


def select_tables(tables, bits):
    """Table 2i + bit_i for every bit (hash_encoding_wtmk_bit.py:110)."""
    if len(tables) != 2 * len(bits):
        raise ValueError(f"codebook has {len(tables)} tables but the message has {len(bits)} bits (expected {len(tables) // 2})")
    return [tables[2 * i + b] for i, b in enumerate(bits)]


def _check_table(t, name):
    if t.dtype != torch.float32 or tuple(t.shape) != (T_ROWS, 2):
        raise ValueError(f"{name}: expected a float32 [{T_ROWS}, 2] table, got {t.dtype} {tuple(t.shape)}")
    return t


def codebook_presum(selected, out=None):
    """S = sum of the D selected codebook tables -> hg_codebook_presum."""
    for i, t in enumerate(selected):
        _check_table(t, f"codebook table {i}")
    S = out if out is not None else torch.empty(T_ROWS, 2, dtype=torch.float32, device=selected[0].device)
    nv.call("hg_codebook_presum", nv.ptr_array([t.detach() for t in selected]), len(selected), nv.ptr(S), nv.stream())
    return S


def codebook_presum_sel(all_tables, message_dev, out=None):
    """S = sum_i table[2i + message[i]] with the selection made on the device (hg_codebook_presum_sel): nothing about the
    call depends on the message's value, so it can sit inside a captured graph."""
    D = len(all_tables) // 2
    if message_dev.dtype != torch.float32 or message_dev.numel() != D or not message_dev.is_cuda:
        raise ValueError(f"message must be a CUDA float32 tensor of {D} entries")
    S = out if out is not None else torch.empty(T_ROWS, 2, dtype=torch.float32, device=message_dev.device)
    nv.call("hg_codebook_presum_sel", nv.ptr_array([_check_table(t.detach(), "codebook table") for t in all_tables]), nv.ptr(message_dev), D,
            nv.ptr(S), nv.stream())
    return S


def encode(x01, base_tables, S=None):
    """[M,32] features of the base encoder (+ codebook through S) -> hg_encode_fwd."""
    x01 = x01.contiguous().float()
    M = x01.shape[0]
    feat = torch.empty(M, 32, dtype=torch.float32, device=x01.device)
    nv.call("hg_encode_fwd", nv.ptr(x01), M, nv.ptr_array([_check_table(t.detach(), "base table") for t in base_tables]), nv.ptr(S),
            nv.ptr(feat), nv.stream())
    return feat


def codebook_encode_literal(x01, selected):
    """[M,2] codebook feature by D separate gathers -> hg_codebook_encode_fwd."""
    x01 = x01.contiguous().float()
    M = x01.shape[0]
    out = torch.empty(M, 2, dtype=torch.float32, device=x01.device)
    nv.call("hg_codebook_encode_fwd", nv.ptr(x01), M, nv.ptr_array([_check_table(t.detach(), "codebook table") for t in selected]),
            len(selected), nv.ptr(out), nv.stream())
    return out


def codebook_scatter(x01, dfeat, G):
    """G += scatter of dfeat [M,2] through the codebook lookup -> hg_codebook_bwd."""
    x01 = x01.contiguous().float()
    dfeat = dfeat.contiguous().float()
    nv.call("hg_codebook_bwd", nv.ptr(x01), x01.shape[0], nv.ptr(dfeat), nv.ptr(G), nv.stream())
    return G


def fanout_grad(G, grads, accumulate=False):
    """grads[i] (+)= G for every selected table -> hg_fanout_grad."""
    nv.call("hg_fanout_grad", nv.ptr(G), nv.ptr_array(grads), len(grads), int(bool(accumulate)), nv.stream())


def level_lookup(x01, resolution):
    x01 = x01.contiguous().float()
    M = x01.shape[0]
    rows = torch.empty(M, 8, dtype=torch.int32, device=x01.device)
    w = torch.empty(M, 3, dtype=torch.float32, device=x01.device)
    nv.call("hg_level_lookup", nv.ptr(x01), M, float(resolution), nv.ptr(rows), nv.ptr(w), nv.stream())
    return rows, w


def pack_weights(sigma_params, color_params, out=None):
    """Split-bf16 MFMA operand image of the two flat parameter vectors -> mlp_pack_weights (into `out` when given)."""
    if sigma_params.numel() != 3072 or color_params.numel() != 7168:
        raise ValueError(f"expected sigma_params[3072] and color_params[7168], got {sigma_params.numel()} and {color_params.numel()}")
    sp = sigma_params.detach().contiguous().float()
    cp = color_params.detach().contiguous().float()
    n = int(nv.fn("mlp_packed_bytes")())
    packed = out if out is not None and out.numel() == n and out.device == sp.device else torch.empty(n, dtype=torch.uint8, device=sp.device)
    nv.call("mlp_pack_weights", nv.ptr(sp), nv.ptr(cp), nv.ptr(packed), nv.stream())
    return packed


PLANES_MIN_POINTS = 16384      # below this the fused kernel wins (one launch, no feature round trip)
PLANES_F32, PLANES_MIXED = 0, 1      # include/nerfsig.h NSIG_PLANES_*: the layout a plane set was written in; its owner passes it to whatever reads the set


def mixed_planes():
    """The feature planes between encoder and MLP in the mixed layout (levels 0..14 as the fp16 pairs the fp16 MLP's first layer consumes, hg_encode_planes_mixed:
    76 instead of 136 bytes per point each way, bit-identical results) -- with the default (fp16) MLP; fp32 planes for the split-bf16 one.  NERFSIG_HALF_PLANES=0: always fp32."""
    return nv.fn("mlp_get_precision")() == 1 and os.environ.get("NERFSIG_HALF_PLANES", "1") != "0"


def encode_planes(xyzs, M, bound, base_ptrs, S, planes, rows_dev=None):
    """The 16 base levels (+ the codebook level through S) of M points into a plane set, in the layout the MLP that follows will read (mixed_planes).
    Returns that layout (PLANES_F32 | PLANES_MIXED): the caller hands it to the launches that read the set."""
    if mixed_planes():
        nv.call("hg_encode_planes_mixed", nv.ptr(xyzs), M, nv.ptr(rows_dev), float(bound), base_ptrs, nv.ptr(S), nv.ptr(planes), nv.stream())
        return PLANES_MIXED
    if rows_dev is not None:
        nv.call("hg_encode_planes_rows", nv.ptr(xyzs), M, nv.ptr(rows_dev), float(bound), base_ptrs, nv.ptr(S), nv.ptr(planes), nv.stream())
    else:
        nv.call("hg_encode_planes", nv.ptr(xyzs), M, float(bound), base_ptrs, nv.ptr(S), nv.ptr(planes), nv.stream())
    return PLANES_F32


def field_forward(xyzs, dirs, bound, base_tables, S, packed, want_rgb=True, want_geo=False, want_masks=False, planes=None, fixed=None):
    """sigma [M], rgb [M,3] | None, geo_feat [M,15] | None, masks | None -> field_fwd.
    planes: True/False forces the two-kernel (XCD-partitioned encoder + MLP) / fused route; None picks by size.
    fixed: a FixedPoints built from these very points -- its base planes are reused, only the codebook level is gathered."""
    xyzs = xyzs.contiguous().float()
    M, dev = xyzs.shape[0], xyzs.device
    sigmas = torch.empty(M, dtype=torch.float32, device=dev)
    rgbs = torch.empty(M, 3, dtype=torch.float32, device=dev) if want_rgb else None
    geo = torch.empty(M, 15, dtype=torch.float32, device=dev) if want_geo else None
    masks = torch.empty((M + 31) // 32 * 32, MASK_WORDS, dtype=torch.int32, device=dev) if want_masks else None
    if want_rgb:
        dirs = dirs.contiguous().float()
    use_planes = (M >= PLANES_MIN_POINTS) if planes is None else bool(planes)
    base_ptrs = nv.ptr_array([_check_table(t.detach(), "base table") for t in base_tables])
    ws, layout = None, PLANES_F32
    if fixed is not None:
        fixed.check(xyzs, bound, base_tables)
        ws, layout = fixed.planes, fixed.layout
        reset = nv.ptr(fixed.plan.buf) if want_masks else None
        if S is not None:      # (a clean render -- no message -- reads the base planes only)
            nv.call("hg_encode_codebook_plane", nv.ptr(xyzs), M, float(bound), nv.ptr(S), nv.ptr(ws), layout, reset, nv.stream())
    elif use_planes:
        ws = torch.empty(int(nv.fn("hg_planes_bytes")(M)), dtype=torch.uint8, device=dev)
        layout = encode_planes(xyzs, M, bound, base_ptrs, S, ws)
    nv.call("field_fwd", nv.ptr(xyzs), nv.ptr(dirs) if want_rgb else None, M, float(bound), base_ptrs, nv.ptr(S), nv.ptr(packed),
            nv.ptr(sigmas), nv.ptr(rgbs), nv.ptr(geo), nv.ptr(masks), nv.ptr(ws), layout, nv.stream())
    return sigmas, rgbs, geo, masks


def field_color(dirs, geo_feat, packed):
    dirs = dirs.contiguous().float()
    geo_feat = geo_feat.contiguous().float()
    M = dirs.shape[0]
    rgbs = torch.empty(M, 3, dtype=torch.float32, device=dirs.device)
    nv.call("field_color_fwd", nv.ptr(dirs), nv.ptr(geo_feat), M, nv.ptr(packed), nv.ptr(rgbs), nv.stream())
    return rgbs


def field_backward(xyzs, bound, g_sigma, g_rgb, sigmas, rgbs, masks, packed, G=None, want_dfeat=False, want_rec=False):
    """Input-gradient backward of the field network.  G: direct scatter (global atomics); want_dfeat: [M,2] d feature[30:32];
    want_rec: the [M,8] scatter record for codebook_scatter_sliced (32 bytes per point: cell, weights, gradients).
    Returns dfeat, rec or (dfeat, rec)."""
    xyzs = xyzs.contiguous().float()
    M = xyzs.shape[0]
    dfeat = torch.empty(M, 2, dtype=torch.float32, device=xyzs.device) if want_dfeat else None
    rec = torch.empty(M, 8, dtype=torch.float32, device=xyzs.device) if want_rec else None
    nv.call("field_bwd", nv.ptr(xyzs), M, float(bound), nv.ptr(g_sigma.contiguous().float()), nv.ptr(g_rgb.contiguous().float()),
            nv.ptr(sigmas), nv.ptr(rgbs), nv.ptr(masks), nv.ptr(packed), nv.ptr(G), nv.ptr(dfeat), nv.ptr(rec), nv.stream())
    if want_dfeat and want_rec:
        return dfeat, rec
    return rec if want_rec else dfeat


BINNED_MIN_POINTS = 65536   # below this the three launches of the binned route cost more than they save


def binned_min_points():
    """Launch size from which the codebook gradient goes through the fixed-point slice owners (bit-reproducible); NERFSIG_DETERMINISTIC=1: every size."""
    from .switches import deterministic
    return 1 if deterministic() else BINNED_MIN_POINTS


def codebook_scatter_sliced(rec, G, binned=None):
    """G += scatter of the [M,8] record emitted by field_backward(want_rec=True).  binned: True -> hg_scatter_binned (hits
    grouped by slice first), False -> hg_scatter_sliced (every owner tests every point), None -> by size."""
    M = rec.shape[0]
    if (M >= binned_min_points()) if binned is None else binned:
        scratch = torch.empty(int(nv.fn("hg_scatter_binned_scratch_bytes")(M)), dtype=torch.uint8, device=rec.device)
        nv.call("hg_scatter_binned", nv.ptr(rec), M, nv.ptr(G), nv.ptr(scratch), nv.stream())
    else:
        nv.call("hg_scatter_sliced", nv.ptr(rec), M, nv.ptr(G), nv.stream())
    return G


_PLAN_STREAM = None
_PENDING_PLANS = []
_LIVE_PLANS = []        # weak references to plans whose `ready` event may still be waited on


def set_plan_stream(stream):
    """Stream on which scatter plans are computed (None: the current stream, immediately).  A plan needs the sample positions
    only; with a plan stream its launches are deferred until flush_plans() -- the caller decides what they queue behind -- or
    until the backward pass asks for the plan, whichever comes first.  Returns the previous setting."""
    global _PLAN_STREAM
    prev, _PLAN_STREAM = _PLAN_STREAM, stream
    return prev


def flush_plans():
    """Enqueue every deferred scatter plan on its plan stream (behind whatever that stream already holds)."""
    while _PENDING_PLANS:
        _PENDING_PLANS.pop(0).launch()


def forget_plan_events():
    """A captured segment ended (trainer.GraphedWatermarkLoop): every launched plan is complete before the next segment starts (the
    segments replay one after the other on one stream), and an event recorded inside a finished capture cannot be waited on in the
    next one -- drop them.  (Deferred plans were flushed by train_step before the streams joined.)"""
    alive = []
    for ref in _LIVE_PLANS:
        plan = ref()
        if plan is not None:
            if plan.launched:
                plan.ready = None
            alive.append(ref)
    _LIVE_PLANS[:] = alive


class ScatterPlan:
    """Counts, offsets and per-point queue destinations of the slice-binned codebook scatter -> hg_scatter_plan."""

    def __init__(self, xyzs, bound):
        self.M, self.xyzs, self.bound = xyzs.shape[0], xyzs, float(bound)
        self.buf = torch.empty(int(nv.fn("hg_scatter_plan_bytes")(self.M)), dtype=torch.uint8, device=xyzs.device)
        self.stream, self.ready, self.launched = _PLAN_STREAM, None, False
        self.src_ready = None
        # A render that itself runs on the plan stream (the content render of the overlapped training step) launches its plan IN LINE, between its
        # march and its encoder.  Round 3 tried deferring it like the block render's: beside the block render's encoder
        # the plan's 1024-thread workgroups starve for wave slots (k_plan_count 231 us instead of 8) and hold the content render back until
        # that encoder has finished -- which turns out to be the better schedule: the content render's encoder then runs beside the block
        # render's MLP (different bottlenecks) instead of beside its encoder (the same texture-address path: block encoder 285 -> 312 us, and
        # the content MLP's 150-VGPR waves find no register space, 180 us instead of 15).  Deferred: 1.056-1.063 ms per step, in line:
        # 1.046-1.057 (profiles/r03_content_plan_deferred_ab.txt).
        if self.stream is not None and self.stream == torch.cuda.current_stream():
            self.stream = None
        if self.stream is None:
            self.launch()
        else:
            if self.stream != torch.cuda.current_stream():
                self.src_ready = torch.cuda.Event()      # the positions are produced on the current stream
                self.src_ready.record()
            _PENDING_PLANS.append(self)

    def launch(self):
        if self.launched:
            return
        self.launched = True
        if self.stream is None:
            nv.call("hg_scatter_plan", nv.ptr(self.xyzs), self.M, self.bound, nv.ptr(self.buf), nv.stream())
        else:
            if self.src_ready is not None:
                self.stream.wait_event(self.src_ready)
            nv.call("hg_scatter_plan", nv.ptr(self.xyzs), self.M, self.bound, nv.ptr(self.buf), self.stream.cuda_stream)
            self.xyzs.record_stream(self.stream)
            self.buf.record_stream(self.stream)
            self.ready = torch.cuda.Event()     # the consumer waits for the plan, not for whatever else that stream runs later
            self.ready.record(self.stream)
            import weakref
            _LIVE_PLANS.append(weakref.ref(self))
        self.xyzs = None

    def join(self):
        """Called on the consumer's stream before the plan is read."""
        if not self.launched:                   # nobody flushed: compute it here
            if self in _PENDING_PLANS:
                _PENDING_PLANS.remove(self)
            self.stream = None
            self.launch()
        elif self.ready is not None:
            if self.stream is None or torch.cuda.current_stream() != self.stream:      # (same stream: already in order -- and an event waited on
                torch.cuda.current_stream().wait_event(self.ready)                     #  by the stream that recorded it crashes hipStreamEndCapture)
            self.ready = None


class _KeptPlan:
    """A scatter plan computed once and kept (FixedPoints): same interface as ScatterPlan, nothing to wait for."""

    def __init__(self, M, buf):
        self.M, self.buf, self.launched, self.ready = M, buf, True, None

    def join(self):
        pass


class FixedPoints:
    """What a field pass over points that never change computes identically every step, computed once.

    The watermark-block rays are one pair of tensors per dataset (nerf/provider_wtmk.py:442-494 builds rays_o_block / rays_d_block in
    the dataset's constructor; every train_step receives the same tensors, utils_wtmk_disen.py:588-590), they are marched without
    jitter (perturb=False, :590) through an occupancy grid the watermark stage never updates, and the base hash tables and both MLPs are
    frozen (network_wtmk_tcnn.py:90-95).  So a step changes ONE input of the block render's field pass: the codebook.  Kept here for
    the points `xyzs` (a buffer the owner re-marches in place, never re-allocates):
      planes  the 16 base-level feature planes (hg_encode_planes without a codebook); plane 16 is rewritten every step by
              hg_encode_codebook_plane from the step's pre-summed codebook;
      plan    the slice-binned scatter's counts, offsets and per-point destinations (hg_scatter_plan: positions only).
    Nothing is approximated: the per-step pass runs the same interpolation code on the same inputs, and the MLP, the compositing,
    the backward pass and the scatter are evaluated every step as before.  refresh() recomputes both IN PLACE (a captured graph
    holds the addresses) after the points or the base tables changed; check() refuses foreign points and refreshes by itself when
    a base table's version moved (eager use; a captured replay runs no Python -- its owner calls refresh)."""

    def __init__(self, xyzs, bound, base_tables):
        self.M, self.bound = int(xyzs.shape[0]), float(bound)
        self.xyzs_ptr = xyzs.data_ptr()
        self.planes = torch.empty(int(nv.fn("hg_planes_bytes")(self.M)), dtype=torch.uint8, device=xyzs.device)
        self.plan = _KeptPlan(self.M, torch.empty(int(nv.fn("hg_scatter_plan_bytes")(self.M)), dtype=torch.uint8, device=xyzs.device))
        self.refreshes = 0
        self.refresh(xyzs, base_tables)

    @staticmethod
    def _tables_key(base_tables):
        # (+ the layout the planes are written in: it follows the MLP's precision mode, see encode_planes)
        return tuple((t.data_ptr(), t._version) for t in base_tables) + (mixed_planes(),)

    def refresh(self, xyzs, base_tables):
        if xyzs.data_ptr() != self.xyzs_ptr or xyzs.shape[0] != self.M:
            raise ValueError("FixedPoints.refresh: these are not the points the cache was built for")
        base_ptrs = nv.ptr_array([_check_table(t.detach(), "base table") for t in base_tables])
        self.layout = encode_planes(xyzs, self.M, self.bound, base_ptrs, None, self.planes)
        nv.call("hg_scatter_plan", nv.ptr(xyzs), self.M, self.bound, nv.ptr(self.plan.buf), nv.stream())
        self.key = self._tables_key(base_tables)
        self.refreshes += 1

    def check(self, xyzs, bound, base_tables):
        if xyzs.data_ptr() != self.xyzs_ptr or xyzs.shape[0] != self.M or float(bound) != self.bound:
            raise ValueError("FixedPoints: the field pass was handed other points than the cache was built for")
        if self._tables_key(base_tables) != self.key and not torch.cuda.is_current_stream_capturing():
            self.refresh(xyzs, base_tables)


def field_backward_planned(xyzs, bound, g_sigma, g_rgb, sigmas, rgbs, masks, packed, plan, G):
    """MLP input gradients written straight into the planned queue, then the slice owners -> field_bwd_planned, hg_scatter_planned."""
    plan.join()
    nv.call("field_bwd_planned", nv.ptr(xyzs), plan.M, float(bound), nv.ptr(g_sigma.contiguous().float()), nv.ptr(g_rgb.contiguous().float()),
            nv.ptr(sigmas), nv.ptr(rgbs), nv.ptr(masks), nv.ptr(packed), nv.ptr(plan.buf), nv.stream())
    nv.call("hg_scatter_planned", nv.ptr(plan.buf), plan.M, nv.ptr(G), nv.stream())
    return G


def field_backward_into(xyzs, bound, g_sigma, g_rgb, sigmas, rgbs, masks, packed, G, plan=None):
    """The production backward: MLP input gradients -> owner-computes scatter into G (through the planned queue when the
    forward pass prepared one, through the 32-byte scatter record otherwise)."""
    if plan is not None:
        return field_backward_planned(xyzs, bound, g_sigma, g_rgb, sigmas, rgbs, masks, packed, plan, G)
    rec = field_backward(xyzs, bound, g_sigma, g_rgb, sigmas, rgbs, masks, packed, want_rec=True)
    return codebook_scatter_sliced(rec, G)


class GradSink:
    """Where the shared codebook gradient G [T,2] of one optimisation step accumulates.

    Every selected table receives the same gradient (see csrc/hashgrid.hip), so all renders of a step
    scatter into one 4 MiB buffer instead of D of them.  The trainer then (optionally) all-reduces G alone
    across data-parallel ranks -- 4 MiB instead of D x 4 MiB -- and either fans it out into the `.grad` of
    the selected tables (`fanout`) or feeds it straight to the fused codebook optimiser."""

    def __init__(self, device, tail=0):
        # `tail` extra floats behind G in the same allocation: room for gradients that travel in the same collective
        # (the decoder's flat gradient block, hidden_models.set_grad_arena)
        self.arena = torch.zeros(T_ROWS * 2 + tail, dtype=torch.float32, device=device)
        self.G = self.arena[:T_ROWS * 2].view(T_ROWS, 2)
        self.tail = self.arena[T_ROWS * 2:]
        self.selected = None  # the parameters the pending gradient belongs to

    def zero_(self):
        self.G.zero_()
        self.selected = None

    def fanout(self, accumulate=False):
        """Materialise `.grad` of every selected table from G (what autograd would have produced)."""
        if self.selected is None:
            return
        slab = torch.empty(len(self.selected), T_ROWS, 2, dtype=torch.float32, device=self.G.device)
        grads = [slab[i] for i in range(len(self.selected))]
        fanout_grad(self.G, grads)
        for p, g in zip(self.selected, grads):
            p.grad = g if (p.grad is None or not accumulate) else p.grad + g


class SharedGradient(GradSink):
    """The sink for a model driven by SOMEBODY ELSE'S training loop (the reference's Trainer under the drop-in modules): it keeps
    `zero_grad() / backward() / [GradScaler.unscale_] / optimizer.step()` meaning what they mean, without ever materialising the D identical dense
    gradients.

    Every selected table's gradient is the same tensor G, so G is handed to autograd's consumers as the `.grad` of ONE of the selected tables (the
    "carrier": GradScaler unscales and inf-checks it like any gradient, exactly once) while the other D - 1 selected tables keep `.grad = None`; a
    global optimiser pre-step hook (optim.install_shared_gradient_hook) then updates all D selected tables from G with the fused Adam pass -- torch.optim.Adam's
    arithmetic and state format -- and clears the carrier's `.grad`, so the optimiser's own loop skips the tables.  Whatever the hook cannot serve (another
    optimiser class, weight decay, a second message inside one accumulation) is handed back to plain autograd semantics by dissolve(): G fanned out
    into real per-table gradients.

    Accumulation across the backward passes of one step: the first backward after a zero_grad() finds the carrier's `.grad` gone (set_to_none) or zeroed in
    place (it IS G) and starts / continues from zero; later ones add."""

    def __init__(self, device):
        super().__init__(device)
        self.carrier = None
        self.live = None          # the selected tables the pending G belongs to
        # the optimiser hook that serves this sink also performs the rest of the (plain Adam) step -- the decoder's dense gradients -- in one launch
        # (optim._dense_takeover); NERFSIG_DROPIN_OFF=dense_adam leaves them to the optimiser's own multi-tensor loop
        from .switches import dropin_off
        self.dense_takeover = not dropin_off("dense_adam")

    def pending(self):
        return self.carrier is not None and self.carrier.grad is self.G

    def begin_accumulation(self, selected):
        """Called by the field backward right before it scatters into G."""
        if self.pending():
            if len(selected) == len(self.live) and all(a is b for a, b in zip(selected, self.live)):
                return                      # same message: the renders of one step add up
            self.dissolve()                 # another message inside one accumulation: G is no longer one gradient for one set of tables
        self.G.zero_()
        self.live = list(selected)
        self.carrier = selected[0]
        if self.carrier.grad is not None:   # an ordinary gradient is already there (fan-out mode earlier in this accumulation): fold it in
            self.G.add_(self.carrier.grad)
        self.carrier.grad = self.G

    def dissolve(self):
        """Back to plain autograd semantics: every selected table gets its own dense gradient (carrier included), G is released."""
        if not self.pending():
            self.carrier = self.live = None
            return
        slab = torch.empty(len(self.live), T_ROWS, 2, dtype=torch.float32, device=self.G.device)
        grads = [slab[i] for i in range(len(self.live))]
        fanout_grad(self.G, grads)
        for p, g in zip(self.live, grads):
            p.grad = g if (p.grad is None or p.grad is self.G) else p.grad + g
        self.carrier = self.live = None

    def consumed(self):
        if self.carrier is not None and self.carrier.grad is self.G:
            self.carrier.grad = None
        self.carrier = self.live = None


class _Tables:
    """The table lists of one field pass, handed to _FieldFunction as ONE opaque argument: autograd (and autocast's argument casting, which walks every
    tensor argument) then tracks only the tables that need it."""
    __slots__ = ("base", "sel")

    def __init__(self, base, sel):
        self.base, self.sel = list(base), list(sel)


class _FieldFunction(Function):
    """NeRFNetwork.forward as one autograd node (network_wtmk_tcnn.py:97-124).

    tabs: the 16 frozen base tables and the D selected codebook tables (_Tables).  `diff` are the autograd inputs among them: without a sink every selected
    table -- backward returns the fan-out of the shared gradient for each (unselected tables are not inputs, so their grad stays None exactly as in the
    reference); with a sink the gradient accumulates there, autograd sees None, and ONE selected table is passed only so that the node is recorded."""

    @staticmethod
    @_fwd32
    def forward(ctx, xyzs, dirs, bound, packed, S, sink, tabs, fixed, *diff):
        base, sel = tabs.base, tabs.sel
        n_sel = len(sel)
        need_grad = n_sel > 0 and len(diff) > 0
        xyzs = xyzs.contiguous().float()
        if fixed is not None:    # points that never change: base planes and scatter plan are kept (FixedPoints)
            ctx.plan = fixed.plan if need_grad else None
        else:
            # before the encoder is enqueued: a plan on its own stream forks right behind the march, not behind this forward pass
            ctx.plan = ScatterPlan(xyzs, bound) if need_grad and xyzs.shape[0] >= binned_min_points() else None
        sigmas, rgbs, _, masks = field_forward(xyzs, dirs, bound, base, S, packed, want_masks=need_grad, fixed=fixed)
        ctx.bound, ctx.n_diff, ctx.need_grad, ctx.sink = bound, len(diff), need_grad, sink
        if need_grad:
            # the saved ReLU masks are laid out for the arithmetic the forward ran in (csrc/field.hip mask_bit<P>): the backward must run in the same
            ctx.mlp_precision = nv.fn("mlp_get_precision")()
            ctx.save_for_backward(xyzs, sigmas, rgbs, masks, packed)
            if sink is not None:
                sink.selected = list(sel)
                ctx.sink_selected = sink.selected
        return sigmas, rgbs

    @staticmethod
    @_bwd
    def backward(ctx, g_sigma, g_rgb):
        head = (None,) * 8
        if not ctx.need_grad:
            return head + (None,) * ctx.n_diff
        xyzs, sigmas, rgbs, masks, packed = ctx.saved_tensors
        plan, ctx.plan = ctx.plan, None
        if nv.fn("mlp_get_precision")() != ctx.mlp_precision:
            raise RuntimeError("mlp_set_precision was called between a field forward pass and its backward: the saved ReLU masks belong to the forward's "
                               "arithmetic (run the backward before switching, or switch before the forward)")
        if ctx.sink is not None:
            if isinstance(ctx.sink, SharedGradient):
                ctx.sink.begin_accumulation(ctx.sink_selected)
            field_backward_into(xyzs, ctx.bound, g_sigma, g_rgb, sigmas, rgbs, masks, packed, ctx.sink.G, plan)
            return head + (None,) * ctx.n_diff
        G = torch.zeros(T_ROWS, 2, dtype=torch.float32, device=xyzs.device)
        field_backward_into(xyzs, ctx.bound, g_sigma, g_rgb, sigmas, rgbs, masks, packed, G, plan)
        slab = torch.empty(ctx.n_diff, T_ROWS, 2, dtype=torch.float32, device=xyzs.device)
        grads = [slab[i] for i in range(ctx.n_diff)]
        fanout_grad(G, grads, accumulate=False)
        return head + tuple(grads)


# ---- differentiable stand-alone density() / color() (network_wtmk_tcnn.py:126-176).  The reference reaches them under autograd only from `run`, which this
# repo evaluates as one joint field pass; a caller that differentiates them directly gets the native forward kernels and a backward written with torch
# operators on the same fp32 weights (not a hot path: two small GEMM chains per call).

_SH4 = None


def sh4_basis(d):
    """Real spherical harmonics of degree < 4 in tiny-cuda-nn's component order, [M,3] -> [M,16], as ONE product of the 20 monomials of degree <= 3 with a
    constant matrix (the kernels evaluate the same polynomials term by term, csrc/field.hip)."""
    global _SH4
    if _SH4 is None or _SH4.device != d.device:
        c1, c2a, c2b, c2c = 0.4886025119029199, 1.0925484305920792, 0.31539156525252005, 0.5462742152960396
        c3a, c3b, c3c, c3d, c3e = 0.5900435899266435, 2.890611442640554, 0.4570457994644658, 0.3731763325901154, 1.445305721320277
        names = ["1", "x", "y", "z", "xy", "yz", "xz", "xx", "yy", "zz", "xxx", "xyy", "xzz", "yxx", "yyy", "yzz", "zxx", "zyy", "zzz", "xyz"]
        rows = {n: i for i, n in enumerate(names)}
        terms = [{"1": 0.28209479177387814}, {"y": -c1}, {"z": c1}, {"x": -c1},
                 {"xy": c2a}, {"yz": -c2a}, {"zz": 2 * c2b, "xx": -c2b, "yy": -c2b}, {"xz": -c2a}, {"xx": c2c, "yy": -c2c},
                 {"yxx": -3 * c3a, "yyy": c3a}, {"xyz": c3b}, {"yzz": -4 * c3c, "yxx": c3c, "yyy": c3c}, {"zzz": 2 * c3d, "zxx": -3 * c3d, "zyy": -3 * c3d},
                 {"xzz": -4 * c3c, "xxx": c3c, "xyy": c3c}, {"zxx": c3e, "zyy": -c3e}, {"xxx": -c3a, "xyy": 3 * c3a}]
        C = torch.zeros(20, 16, dtype=torch.float64)
        for k, t in enumerate(terms):
            for n, v in t.items():
                C[rows[n], k] = v
        _SH4 = C.float().to(d.device)
    x, y, z = d.unbind(-1)
    xx, yy, zz = x * x, y * y, z * z
    mono = torch.stack([torch.ones_like(x), x, y, z, x * y, y * z, x * z, xx, yy, zz, x * xx, x * yy, x * zz, y * xx, y * yy, y * zz, z * xx, z * yy, z * zz, x * y * z], dim=-1)
    return mono @ _SH4


class _DensityFunction(Function):
    """NeRFNetwork.density (network_wtmk_tcnn.py:126-144) as an autograd node towards the selected codebook tables: (sigma [M], geo_feat [M,15])."""

    @staticmethod
    @_fwd32
    def forward(ctx, xyzs, bound, packed, S, sink, tabs, sigma_params, *diff):
        xyzs = xyzs.contiguous().float()
        sigmas, _, geo, _ = field_forward(xyzs, None, bound, tabs.base, S, packed, want_rgb=False, want_geo=True)
        ctx.bound, ctx.n_diff, ctx.sink, ctx.tabs, ctx.S = float(bound), len(diff), sink, tabs, S
        ctx.save_for_backward(xyzs, sigmas, sigma_params.detach())
        if sink is not None:
            sink.selected = list(tabs.sel)
            ctx.sink_selected = sink.selected
        return sigmas, geo

    @staticmethod
    @_bwd
    def backward(ctx, g_sigma, g_geo):
        head = (None,) * 7
        xyzs, sigmas, params = ctx.saved_tensors
        M, dev = xyzs.shape[0], xyzs.device
        W1, W2 = params[:2048].view(64, 32), params[2048:3072].view(16, 64)          # tcnn layout: [out][in] row-major
        x01 = (xyzs + ctx.bound) / (2 * ctx.bound)
        feat = encode(x01, ctx.tabs.base, ctx.S)                                     # the 32 encoder features, codebook included
        pre = feat @ W1.t()
        d_so = torch.zeros(M, 16, dtype=torch.float32, device=dev)
        if g_sigma is not None:      # trunc_exp's backward: g * exp(clamp(h0, -15, 15)) (activation.py:11-15); sigma = exp(h0)
            d_so[:, 0] = g_sigma.float() * sigmas.clamp(min=3.059023205018258e-07, max=3269017.3724721107)
        if g_geo is not None:
            d_so[:, 1:] = g_geo.float()
        dfeat = (((d_so @ W2) * (pre > 0)) @ W1)[:, 30:32].contiguous()               # the codebook is added into features 30:32 (network_wtmk_tcnn.py:106)
        if ctx.sink is not None:
            if isinstance(ctx.sink, SharedGradient):
                ctx.sink.begin_accumulation(ctx.sink_selected)
            codebook_scatter(x01, dfeat, ctx.sink.G)
            return head + (None,) * ctx.n_diff
        G = torch.zeros(T_ROWS, 2, dtype=torch.float32, device=dev)
        codebook_scatter(x01, dfeat, G)
        slab = torch.empty(ctx.n_diff, T_ROWS, 2, dtype=torch.float32, device=dev)
        grads = [slab[i] for i in range(ctx.n_diff)]
        fanout_grad(G, grads, accumulate=False)
        return head + tuple(grads)


def density_apply(xyzs, bound, packed, base_tables, selected, S, sink, sigma_params):
    """(sigma, geo_feat) with autograd to the selected codebook tables (density() under gradients)."""
    if sink is None:
        diff = tuple(selected) if any(t.requires_grad for t in selected) else ()
    else:
        diff = next(((t,) for t in selected if t.requires_grad), ())
    return _DensityFunction.apply(xyzs, bound, packed, S, sink, _Tables(base_tables, selected), sigma_params, *diff)


class _ColorFunction(Function):
    """NeRFNetwork.color without the mask (network_wtmk_tcnn.py:147-176) as an autograd node towards geo_feat (the colour MLP is frozen in the watermark stage)."""

    @staticmethod
    @_fwd32
    def forward(ctx, dirs, geo_feat, packed, color_params):
        rgbs = field_color(dirs, geo_feat, packed)
        ctx.save_for_backward(dirs.contiguous().float(), geo_feat.contiguous().float(), rgbs, color_params.detach())
        return rgbs

    @staticmethod
    @_bwd
    def backward(ctx, g_rgb):
        dirs, geo, rgbs, params = ctx.saved_tensors
        Wc1, Wc2, Wc3 = params[:2048].view(64, 32), params[2048:6144].view(64, 64), params[6144:7168].view(16, 64)
        cin = torch.cat([sh4_basis(((dirs + 1) / 2) * 2 - 1), geo, torch.ones_like(geo[:, :1])], dim=-1)      # (:166-171: SH of the direction mapped to [0, 1] and back)
        p1 = cin @ Wc1.t()
        p2 = torch.relu(p1) @ Wc2.t()
        d_out = torch.zeros(dirs.shape[0], 16, dtype=torch.float32, device=dirs.device)
        d_out[:, :3] = g_rgb.float() * rgbs * (1 - rgbs)
        d_cin = (((d_out @ Wc3) * (p2 > 0)) @ Wc2 * (p1 > 0)) @ Wc1
        return None, d_cin[:, 16:31].contiguous(), None, None


def color_apply(dirs, geo_feat, packed, color_params):
    return _ColorFunction.apply(dirs, geo_feat, packed, color_params)


def field_apply(xyzs, dirs, bound, packed, base_tables, selected, S=None, sink=None, fixed=None):
    """(sigma, rgb) with autograd to the selected codebook tables.  S: their pre-sum (computed here if omitted).
    fixed: the FixedPoints of exactly these points (rays that do not change between steps), or None."""
    if len(selected) and S is None:
        S = codebook_presum(selected)
    if not torch.is_grad_enabled():
        diff = ()
    elif sink is None:
        # plain autograd: every selected table is an input (a table that does not require grad gets a gradient autograd drops)
        diff = tuple(selected) if any(t.requires_grad for t in selected) else ()
    else:           # the gradient goes to the sink: one differentiable input is enough to have the node recorded
        diff = next(((t,) for t in selected if t.requires_grad), ())
    return _FieldFunction.apply(xyzs, dirs, bound, packed, S, sink, _Tables(base_tables, selected), fixed, *diff)
