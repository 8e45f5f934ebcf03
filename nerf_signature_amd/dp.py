"""Data parallelism of the watermark step: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on
ROCm; "gloo" for the CPU tests and for rehearsing N ranks on a one-GPU box).

Partitioning.  Rays shard: rank r draws its own content rays, and the D watermark blocks are split
too -- rank r renders blocks [r*D/R, (r+1)*D/R).  The HiDDeN decoder's BatchNorm uses batch statistics over the D blocks
(hidden_models.py:26), so the decoder itself is replicated: the rendered blocks ([D,bh,bw,3], 55 KB at hotdog) are
ALL-GATHERED, every rank decodes all D of them and back-propagates the image gradient of its own blocks only.

Two collectives per step:
  1. all-gather of the rendered blocks (gather_blocks), in the middle of the forward pass;
  2. one all-reduce (SUM) of [G | decoder gradients] (GradExchange).  Every selected codebook table receives the same
     gradient G [T,2] (csrc/hashgrid.hip), so ranks reduce G alone -- 4 MiB -- instead of D dense gradients (128 MiB at D=32).
     With sharded blocks the block parts of G add up (sum) and the content parts are averaged: the step seeds the content
     loss's backward with 1/R (content_grad_scale), so a plain SUM yields  sum_r G_block_r + mean_r G_content_r, the
     single-process gradient of  lambda_w * BCE(all D blocks) + lambda_i * MSE(all R*4096 content rays).  The decoder's
     gradients are identical on every rank (same gathered images); their sum is divided by R (keeps replicas in lockstep).
Without sharding (D not divisible by R, or NERFSIG_REPLICATE_BLOCKS=1) the block render is replicated and everything is
averaged, as in round 1.

Inside a captured step a collective is a SEGMENT BOUNDARY: `collective(fn)` hands `fn` to the capturing loop, which ends the
running graph capture, remembers `fn` to be executed eagerly at that point of every replay, and begins the next segment."""
import os

import torch
import torch.distributed as dist

_BOUNDARY = None      # installed by a capturing loop (trainer.GraphedWatermarkLoop): callable(fn)


def set_boundary(handler):
    """handler(fn): called instead of fn() when a collective is reached while the current stream is capturing.  Returns the previous one."""
    global _BOUNDARY
    prev, _BOUNDARY = _BOUNDARY, handler
    return prev


_TIMER = None         # name -> [(start, end)] while a measurement pass is on (time_collectives)


def time_collectives(on=True):
    """Bracket every eagerly executed collective with events on the stream it is issued from (HIP events; wall clock for CPU groups) from now on / stop.
    A collective that is captured inside a hipGraph (NERFSIG_CAPTURE_COLLECTIVES=1) cannot be bracketed: bench.py times those in its eager pass."""
    global _TIMER
    _TIMER = {} if on else None


def collective_times_us():
    """name -> {"n", "mean", "min", "max"} in microseconds, of what time_collectives() has seen (synchronises).  The times include the wait for the slowest
    rank to arrive -- they are what the step pays, not the wire time."""
    out = {}
    if _TIMER is None:
        return out
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    for name, ev in _TIMER.items():
        us = [(a.elapsed_time(b) * 1e3 if hasattr(a, "elapsed_time") else (b - a) * 1e6) for a, b in ev]
        if us:
            out[name] = {"n": len(us), "mean": sum(us) / len(us), "min": min(us), "max": max(us)}
    return out


def _timed(name, fn):
    def run():
        t = _TIMER
        if t is None or (torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()):
            return fn()
        if torch.cuda.is_available() and dist.is_initialized() and dist.get_backend() != "gloo":
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            fn()
            b.record()
        else:
            import time
            a = time.perf_counter()
            fn()
            b = time.perf_counter()
        t.setdefault(name, []).append((a, b))
    return run


def collective(fn, last=False, name="collective"):
    """Run the eager collective(s) in `fn` now -- or, under a segmented capture, make this point a segment boundary.
    NERFSIG_CAPTURE_COLLECTIVES=1 leaves the RCCL calls INSIDE the capture instead: one hipGraph per step for any world size, no eager launches
    between segments (-85 us per step on a world-size-1 nccl group).  The capture then must not start while ProcessGroupNCCL's watchdog thread still
    holds work of earlier eager collectives (drain_watchdog; LABNOTES section 16).  name: the key under which time_collectives() files its duration."""
    fn = _timed(name, fn)
    if _BOUNDARY is not None and torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
        if os.environ.get("NERFSIG_CAPTURE_COLLECTIVES", "") == "1":
            fn()
        else:
            _BOUNDARY(fn, last)        # last: nothing of the step follows this collective -- no further segment is opened
    else:
        fn()


def drain_watchdog(limit_s=5.0):
    """Block until ProcessGroupNCCL's watchdog thread holds no work of earlier (eager) collectives.  Call after torch.cuda.synchronize() and before a graph capture.

    Why: the watchdog polls the end event of every eager collective (hipEventQuery, every 100 ms) until it has seen it complete, then retires the work.  One such
    query from that thread while THIS thread was capturing killed a rehearsal in round 2 (WorkNCCL::isCompleted raising inside Watchdog::runLoop: the process
    aborts).  Collectives issued under capture are never handed to the watchdog, so the only works it can hold are those of the warm-up steps -- all complete
    after the synchronize; what remains is for the watchdog to notice.  Rounds 2-4 slept a fixed 0.5 s (5 polling periods).  Here the condition itself is polled:
    the flight recorder marks an entry `retired` when the watchdog drops its work, and _dump_nccl_trace(onlyActive=True) lists the others; empty list = empty
    watchdog.  Falls back to the fixed sleep where the recorder is off (TORCH_NCCL_TRACE_BUFFER_SIZE=0) or the call is missing.  Returns what it did (for the record)."""
    global LAST_DRAIN
    LAST_DRAIN = _drain_watchdog(limit_s)
    return LAST_DRAIN


LAST_DRAIN = None     # what the most recent drain_watchdog() did (bench.py prints it)


def _drain_watchdog(limit_s):
    import time
    t0 = time.perf_counter()
    if not (dist.is_initialized() and dist.get_backend() == "nccl"):
        return {"how": "no nccl group", "seconds": 0.0}
    dump = getattr(torch._C._distributed_c10d, "_dump_nccl_trace", None)
    seen_any = False
    if dump is not None:
        import pickle
        try:
            while time.perf_counter() - t0 < limit_s:
                doc = pickle.loads(dump(includeCollectives=True, includeStackTraces=False, onlyActive=True))
                entries = doc.get("entries", []) if isinstance(doc, dict) else []
                if not entries:
                    if not seen_any:
                        # nothing listed on the first look: either the watchdog is already empty or the recorder is off -- tell the two apart
                        full = pickle.loads(dump(includeCollectives=True, includeStackTraces=False, onlyActive=False))
                        if not (isinstance(full, dict) and full.get("entries")):
                            break           # recorder off: fixed sleep below
                    time.sleep(0.12)        # one more polling period: a work is retired in the same pass that finds it complete
                    return {"how": "flight recorder: no active work left", "seconds": time.perf_counter() - t0, "waited_for_entries": seen_any}
                seen_any = True
                time.sleep(0.02)
        except Exception as e:      # noqa: BLE001 -- a private API: any surprise falls back to the sleep
            seen_any = repr(e)
    time.sleep(0.5)
    return {"how": "fixed 0.5 s sleep (flight recorder unavailable)", "seconds": time.perf_counter() - t0, "note": seen_any}


def collective_ends_segment():
    """True when a collective issued now would end the running capture segment (see collective): forked streams have to join in front of it."""
    return (_BOUNDARY is not None and torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()
            and os.environ.get("NERFSIG_CAPTURE_COLLECTIVES", "") != "1")


def init_from_env(backend=None):
    """Initialise the default process group from RANK / WORLD_SIZE / MASTER_* (torchrun).  Returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or os.environ.get("NERFSIG_FORCE_EXCHANGE", "") == "1") and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = os.environ.get("NERFSIG_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local_rank % max(torch.cuda.device_count(), 1))
            os.environ.setdefault("TORCH_NCCL_TRACE_BUFFER_SIZE", "2000")      # the flight recorder drain_watchdog() reads (must be set before the group exists)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local_rank


def device_ordinal(local_rank=None, device_count=None):
    """The device this rank binds (`torch.cuda.set_device(ordinal)`) and the physical GPU behind it, decided from the environment alone -- no GPU
    call: ordinal = LOCAL_RANK modulo the visible device count; physical = that entry of HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES /
    CUDA_VISIBLE_DEVICES when one of them restricts the process, else the ordinal itself.  Returns (ordinal, physical id string)."""
    local_rank = int(os.environ.get("LOCAL_RANK", "0")) if local_rank is None else int(local_rank)
    visible = None
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        if os.environ.get(var, "").strip():
            visible = [v.strip() for v in os.environ[var].split(",") if v.strip()]
            break
    if device_count is None:
        device_count = len(visible) if visible is not None else torch.cuda.device_count()      # (device_count() does not initialise the device)
    ordinal = local_rank % max(int(device_count), 1)
    return ordinal, (visible[ordinal] if visible is not None and ordinal < len(visible) else str(ordinal))


def assert_distinct_devices(device_count=None):
    """Every rank of this node binds a GPU of its own -- checked over the process group's HOST side (all_gather_object of (host, physical id) on a
    gloo group, or per rank from LOCAL_RANK / LOCAL_WORLD_SIZE alone when the group is nccl: object collectives there would touch the GPU) before
    any kernel is launched.  Raises RuntimeError naming the clash.  Returns this rank's (ordinal, physical id)."""
    import socket
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")))
    ordinal, physical = device_ordinal(local_rank, device_count)
    count = device_count if device_count is not None else (torch.cuda.device_count() if not any(os.environ.get(v, "").strip() for v in
                                                           ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES")) else None)
    if count is not None and local_world > count:
        partner = next((r for r in range(local_world) if r != local_rank and r % count == local_rank % count), None)      # (a rank's device is LOCAL_RANK mod count)
        raise RuntimeError(f"{local_world} ranks on this node but {count} visible GPU(s): LOCAL_RANK {local_rank} would share device {ordinal}"
                           + (f" with LOCAL_RANK {partner}" if partner is not None else " (other ranks of the node collide)")
                           + " (NERFSIG_DIST_BACKEND=gloo rehearses more ranks than GPUs)")
    if dist.is_initialized() and dist.get_backend() == "gloo" and dist.get_world_size() > 1:
        mine = (socket.gethostname(), physical)
        everyone = [None] * dist.get_world_size()
        dist.all_gather_object(everyone, mine)
        seen = {}
        for r, key in enumerate(everyone):
            if key in seen:
                raise RuntimeError(f"ranks {seen[key]} and {r} bind the same GPU {key[1]} on {key[0]}")
            seen[key] = r
    return ordinal, physical


def world_size():
    return dist.get_world_size() if dist.is_initialized() else 1


def exchange_active():
    """True when a step has an exchange phase: more than one rank, or NERFSIG_FORCE_EXCHANGE=1 with an initialised process
    group of ANY size -- a one-GPU box can then rehearse the multi-rank execution (two graphs with the RCCL collectives
    between them) through a world-size-1 "nccl" group; the sums over one rank leave every value unchanged."""
    if not dist.is_initialized():
        return False
    return dist.get_world_size() > 1 or os.environ.get("NERFSIG_FORCE_EXCHANGE", "") == "1"


def rank():
    return dist.get_rank() if dist.is_initialized() else 0


def block_shard(D):
    """(first, one-past-last) watermark block of this rank, or None when the block render is replicated (no exchange phase, D not
    divisible by the world size, or NERFSIG_REPLICATE_BLOCKS=1).  A world-size-1 rehearsal group shards into one piece."""
    if not exchange_active() or os.environ.get("NERFSIG_REPLICATE_BLOCKS", "") == "1":
        return None
    world = dist.get_world_size()
    if D % world != 0:
        return None
    n = D // world
    return dist.get_rank() * n, (dist.get_rank() + 1) * n


def optimizer_shard(D):
    """(first, one-past-last) message BIT whose two codebook tables this rank's optimiser owns, or None (every rank updates all tables).
    With the blocks sharded the step never reads a codebook table directly -- both renders read the pre-summed table S, and the
    backward writes the shared gradient G -- so Adam over the D selected tables (836 MiB of HBM traffic, the largest replicated item of a
    rank's step) is split like ZeRO: rank r updates the tables of its bits with the all-reduced G and contributes its PARTIAL pre-sum of
    the next message to one more 4 MiB all-reduce.  Tables a rank does not own go stale there until GraphedWatermarkLoop.gather_codebook().
    It pays from four ranks on: it saves (1 - 1/R) of a ~150 us HBM-bound kernel and costs one more latency-bound collective (~40 us).
    NERFSIG_SHARD_OPTIMIZER=0 / 1 forces it off / on."""
    mode = os.environ.get("NERFSIG_SHARD_OPTIMIZER", "")
    if mode == "0" or (mode != "1" and world_size() < 4 and os.environ.get("NERFSIG_FORCE_EXCHANGE", "") != "1"):
        return None
    return block_shard(D)


def content_grad_scale(sharded):
    """Factor on the content loss's backward seed: 1/R with sharded blocks (the exchange then only sums), else 1."""
    return 1.0 / world_size() if sharded else 1.0


@torch.no_grad()
def _all_gather_into(out, local):
    if dist.get_backend() == "gloo":     # (no all_gather_into_tensor on gloo: chunk views of `out` as the output list)
        dist.all_gather(list(out.chunk(dist.get_world_size(), dim=0)), local)
    else:
        dist.all_gather_into_tensor(out, local)


class _GatherBlocks(torch.autograd.Function):
    """[D/R, ...] rendered blocks of this rank -> [D, ...] of all ranks (rank-major = block order).  Backward: every rank holds the
    full image gradient (the decoder is replicated), so it just keeps the rows of its own blocks -- no communication."""

    @staticmethod
    def forward(ctx, local, D, first):
        local = local.contiguous()
        out = torch.empty((D,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        ctx.rows = (first, local.shape[0])
        dst, src = out.detach(), local.detach()     # aliases outside autograd: a replay runs this call long after `out` became a graph output
        collective(lambda: _all_gather_into(dst, src), name="all_gather_blocks")
        return out

    @staticmethod
    def backward(ctx, g):
        first, n = ctx.rows
        return g[first:first + n].contiguous(), None, None


def gather_blocks(local_image, D, first):
    return _GatherBlocks.apply(local_image, D, first)


def _tiled_range(grads):
    """(storage, first element, one-past-last element) if `grads` are contiguous float32 pieces of ONE storage that tile a
    contiguous element range without gaps or overlaps; else None."""
    if not grads or any(g.dtype != torch.float32 or not g.is_contiguous() for g in grads):
        return None
    st = grads[0].untyped_storage()
    if any(g.untyped_storage().data_ptr() != st.data_ptr() for g in grads):
        return None
    pieces = sorted((g.storage_offset(), g.numel()) for g in grads)
    end = pieces[0][0]
    for off, cnt in pieces:
        if off != end:
            return None
        end += cnt
    return st, pieces[0][0], end


def _common_base(grads):
    """One flat tensor over the storage all `grads` share, if they are contiguous float32 pieces that tile it exactly (the fused
    decoder hands out such pieces; autograd keeps them as `.grad` without copying, though not as registered views)."""
    if not grads or any(g.dtype != torch.float32 or not g.is_contiguous() for g in grads):
        return None
    st = grads[0].untyped_storage()
    if any(g.untyped_storage().data_ptr() != st.data_ptr() for g in grads):
        return None
    n = st.nbytes() // 4
    pieces = sorted((g.storage_offset(), g.numel()) for g in grads)
    end = 0
    for off, cnt in pieces:
        if off != end:
            return None
        end += cnt
    if end != n:
        return None
    return torch.empty(0, dtype=torch.float32, device=grads[0].device).set_(st, 0, (n,))


class GradExchange:
    """All-reduce of (shared codebook gradient, decoder gradients) with the decoder in one flat bucket.

    average=True (default): after the SUM, G is multiplied by `shared_scale` (default 1/world) and the decoder's gradients by
    1/world, in place.  average=False: the plain sum -- the caller folds the factors into the optimiser kernels (`grad_scale` of
    CodebookAdam.step_shared_sel / step_dense), which saves the scaling launches of the captured step.
    shared_scale: 1.0 when the blocks are sharded and the content seeds already carry 1/world (module docstring)."""

    def __init__(self, decoder_params, average=True, shared_scale=None):
        self.average = average
        self.shared_scale = shared_scale
        self.decoder_params = [p for p in decoder_params if p.requires_grad]
        n = sum(p.numel() for p in self.decoder_params)
        p0 = self.decoder_params[0]
        self.bucket = torch.zeros(n, dtype=torch.float32, device=p0.device)
        self.bytes_per_step = 0
        self.collectives_per_step = 2

    def __call__(self, shared_grad):
        """shared_grad: the GradSink's G (or None).  In-place mean over ranks of G and of every decoder .grad."""
        world = world_size()
        if not exchange_active():
            return
        s_shared = (1.0 / world) if self.shared_scale is None else float(self.shared_scale)
        s_dec = 1.0 / world
        grads = [p.grad for p in self.decoder_params if p.grad is not None]
        if shared_grad is not None and shared_grad.is_contiguous() and shared_grad.dtype == torch.float32:
            # G and the decoder's gradient block behind it in one allocation (GradSink(tail=...) + hidden_models.set_grad_arena):
            # ONE collective for the whole step -- on xGMI a 5 MiB all-reduce costs a latency, not bandwidth, so two cost twice
            rng = _tiled_range(grads)
            g0 = shared_grad.storage_offset()
            if rng is not None and rng[0].data_ptr() == shared_grad.untyped_storage().data_ptr() and rng[1] == g0 + shared_grad.numel():
                both = torch.empty(0, dtype=torch.float32, device=shared_grad.device).set_(rng[0], g0, (rng[2] - g0,))
                collective(lambda: dist.all_reduce(both, op=dist.ReduceOp.SUM), name="all_reduce_gradients")
                if self.average:
                    if s_shared == s_dec:
                        both.mul_(s_dec)
                    else:
                        both[:shared_grad.numel()].mul_(s_shared)
                        both[shared_grad.numel():].mul_(s_dec)
                self.bytes_per_step = both.numel() * 4
                self.collectives_per_step = 1
                return
        flat = _common_base(grads)
        if flat is not None:   # the gradients already live in one buffer: one collective, no copies

            def both_reduces():
                hs = [dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=True) for t in (shared_grad, flat) if t is not None]
                for h in hs:
                    h.wait()

            collective(both_reduces, name="all_reduce_gradients")
            if self.average:
                if shared_grad is not None:
                    shared_grad.mul_(s_shared)
                flat.mul_(s_dec)
            self.bytes_per_step = (shared_grad.numel() * 4 if shared_grad is not None else 0) + flat.numel() * 4
            return
        off = 0
        for p in self.decoder_params:
            n = p.numel()
            if p.grad is None:
                self.bucket[off:off + n].zero_()
            else:
                self.bucket[off:off + n].copy_(p.grad.reshape(-1))
            off += n

        def bucket_reduces():
            hs = [dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=True) for t in (shared_grad, self.bucket) if t is not None]
            for h in hs:
                h.wait()

        collective(bucket_reduces, name="all_reduce_gradients")
        inv = s_dec if self.average else 1.0
        if shared_grad is not None and self.average:
            shared_grad.mul_(s_shared)
        off = 0
        for p in self.decoder_params:
            n = p.numel()
            g = self.bucket[off:off + n].view_as(p) * inv
            if p.grad is None:
                p.grad = g.clone()
            else:
                p.grad.copy_(g)
            off += n
        self.bytes_per_step = (shared_grad.numel() * 4 if shared_grad is not None else 0) + self.bucket.numel() * 4


def allreduce_gradients(params, bucket_bytes=64 << 20):
    """Mean all-reduce of the .grad of `params` in flat buckets (stage-1 training: base tables + MLPs, 64 MiB + 40 KB):
    buckets of ~64 MiB keep every xGMI link busy with few, large collectives."""
    world = world_size()
    if world == 1:
        return 0
    grads = [p.grad for p in params if p.grad is not None]
    total, i = 0, 0
    while i < len(grads):
        j, size = i, 0
        while j < len(grads) and (size == 0 or size + grads[j].numel() * 4 <= bucket_bytes):
            size += grads[j].numel() * 4
            j += 1
        flat = torch.cat([g.reshape(-1) for g in grads[i:j]])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat.mul_(1.0 / world)
        off = 0
        for g in grads[i:j]:
            g.copy_(flat[off:off + g.numel()].view_as(g))
            off += g.numel()
        total += size
        i = j
    return total
