"""Data-parallel gradient exchange: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on
ROCm; "gloo" for the CPU tests).

What is exchanged per step: the decoder's gradients (261 893 fp32, one flat bucket) and the
codebook gradient.  Because every selected codebook table receives the same gradient G [T,2]
(csrc/hashgrid.hip), ranks all-reduce G alone -- 4 MiB -- and fan it out locally, instead of all-reducing
D dense [T,2] gradients (128 MiB at D=32).  Rays are sharded by rank (distinct content rays per rank), the
block render and the message are replicated, so the result equals the single-process gradient of the mean loss."""
import os

import torch
import torch.distributed as dist


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
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local_rank


def world_size():
    return dist.get_world_size() if dist.is_initialized() else 1


def exchange_active():
    """True when a step has an exchange phase: more than one rank, or NERFSIG_FORCE_EXCHANGE=1 with an initialised process
    group of ANY size -- a one-GPU box can then rehearse the multi-rank execution (two graphs with the RCCL collectives
    between them) through a world-size-1 "nccl" group; the sums over one rank leave every value unchanged."""
    if not dist.is_initialized():
        return False
    return dist.get_world_size() > 1 or os.environ.get("NERFSIG_FORCE_EXCHANGE", "") == "1"


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

    average=True (default): the mean over ranks, in place.  average=False: the plain sum -- the caller folds 1/world into
    the optimiser kernels (`grad_scale` of CodebookAdam.step_shared_sel / step_dense), which saves the two scaling
    launches between the graphs of the captured step."""

    def __init__(self, decoder_params, average=True):
        self.average = average
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
        grads = [p.grad for p in self.decoder_params if p.grad is not None]
        if shared_grad is not None and shared_grad.is_contiguous() and shared_grad.dtype == torch.float32:
            # G and the decoder's gradient block behind it in one allocation (GradSink(tail=...) + hidden_models.set_grad_arena):
            # ONE collective for the whole step -- on xGMI a 5 MiB all-reduce costs a latency, not bandwidth, so two cost twice
            rng = _tiled_range(grads)
            g0 = shared_grad.storage_offset()
            if rng is not None and rng[0].data_ptr() == shared_grad.untyped_storage().data_ptr() and rng[1] == g0 + shared_grad.numel():
                both = torch.empty(0, dtype=torch.float32, device=shared_grad.device).set_(rng[0], g0, (rng[2] - g0,))
                dist.all_reduce(both, op=dist.ReduceOp.SUM)
                if self.average:
                    both.mul_(1.0 / world)
                self.bytes_per_step = both.numel() * 4
                self.collectives_per_step = 1
                return
        handles = []
        if shared_grad is not None:
            handles.append(dist.all_reduce(shared_grad, op=dist.ReduceOp.SUM, async_op=True))
        flat = _common_base(grads)
        if flat is not None:   # the gradients already live in one buffer: one collective, no copies
            handles.append(dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True))
            for h in handles:
                h.wait()
            if self.average:
                if shared_grad is not None:
                    shared_grad.mul_(1.0 / world)
                flat.mul_(1.0 / world)
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
        handles.append(dist.all_reduce(self.bucket, op=dist.ReduceOp.SUM, async_op=True))
        for h in handles:
            h.wait()
        inv = 1.0 / world if self.average else 1.0
        if shared_grad is not None:
            shared_grad.mul_(inv)
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
