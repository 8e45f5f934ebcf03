"""Two data-parallel ranks of the REAL watermark step emulated inside one process on one GPU (the pool's boxes have one GPU and RCCL
refuses two ranks on one device): each "rank" is a thread with its own model replica running trainer.WatermarkLoop.step unchanged, and
torch.distributed's entry points are replaced by an in-process stand-in that implements all_gather / all_reduce over the two threads.
Everything above the collectives -- block sharding (dp.block_shard), the all-gather in front of the replicated decoder, the 1/world seed
of the content loss, the single SUM all-reduce of [G | decoder gradients] -- is the production code.

Bar: the exchanged gradient equals the single-process gradient of the step on ALL D blocks and the
concatenated content batch of both ranks: relative L2 <= 1e-5 (float atomics reorder sums)."""
import threading

import numpy as np
import pytest
import torch

import closed_form as cf
from test_gpu_render import _data, _model

pytestmark = pytest.mark.gpu
KW = dict(dt_gamma=0, max_steps=1024)


class _Work:
    def wait(self):
        return True


class TwoRanks:
    """Stand-in for the torch.distributed functions nerf_signature_amd.dp uses, over `world` threads of this process."""

    def __init__(self, world=2):
        self.world = world
        self.local = threading.local()
        self.barrier = threading.Barrier(world)
        self.turn = threading.Lock()      # one rank computes at a time (the package keeps per-step state in module globals: one process = one rank)
        self.slots = [None] * world
        self.log = []

    # -- the torch.distributed surface dp.py touches
    def is_initialized(self):
        return True

    def get_world_size(self):
        return self.world

    def get_rank(self):
        return self.local.rank

    def get_backend(self):
        return "nccl"

    def _exchange(self, t):
        self.slots[self.local.rank] = t
        self.turn.release()
        try:
            self.barrier.wait()
            parts = list(self.slots)
            self.barrier.wait()
        finally:
            self.turn.acquire()
        return parts

    def all_gather_into_tensor(self, out, local):
        parts = self._exchange(local.clone())
        out.copy_(torch.cat(parts, dim=0))
        self.log.append(("all_gather", self.local.rank, out.numel() * 4))

    def all_reduce(self, t, op=None, async_op=False):
        parts = self._exchange(t.clone())
        t.copy_(sum(parts[1:], parts[0]))
        self.log.append(("all_reduce", self.local.rank, t.numel() * 4))
        return _Work() if async_op else None


class Recorder:
    """Optimiser stand-in: keeps what the loop hands to the optimiser step (the exchanged gradients) instead of applying it."""

    def __init__(self, model):
        self.model, self.G, self.dec = model, None, None

    def zero_grad(self, set_to_none=True):
        for p in self.model.parameters():
            p.grad = None

    def step_shared(self, selected, G, grad_scale=1.0):
        self.G = G.detach().clone()
        self.n_selected = len(selected)

    def step(self):
        self.dec = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in self.model.msg_decoder.parameters()]).clone()


def _run_single(data, msg):
    from nerf_signature_amd import trainer
    torch.manual_seed(0)             # the decoder's initialisation: identical replicas
    m, _, _ = _model()
    rec = Recorder(m)
    out = trainer.WatermarkLoop(m, rec, KW).step(data, msg)
    torch.cuda.synchronize()
    return rec, [float(v.detach()) for v in out[3:6]]


def test_two_rank_step_equals_single_process_step(monkeypatch, strict_mlp):
    import torch.distributed as dist
    from nerf_signature_amd import dp, trainer
    bo, bd, co0, cd0, gt0 = _data(n_content=300, seed=0)
    _, _, co1, cd1, gt1 = _data(n_content=300, seed=1)
    gt1 = (gt1 * 0.7).contiguous()
    msg = torch.from_numpy(cf.messages(32)[2])
    wm = {"rays_o_block": bo.cuda(), "rays_d_block": bd.cuda()}
    per_rank = [{"watermark": wm, "content": {"rays_o": o.cuda(), "rays_d": d.cuda(), "images": g.cuda()}} for o, d, g in ((co0, cd0, gt0), (co1, cd1, gt1))]
    both = {"watermark": wm, "content": {"rays_o": torch.cat([co0, co1], 1).cuda(), "rays_d": torch.cat([cd0, cd1], 1).cuda(), "images": torch.cat([gt0, gt1], 1).cuda()}}
    assert not torch.equal(co0, co1) or not torch.equal(cd0, cd1)

    ref, ref_losses = _run_single(both, msg)             # one process, all 32 blocks, 600 content rays

    group = TwoRanks(2)
    for name in ("is_initialized", "get_world_size", "get_rank", "get_backend", "all_gather_into_tensor", "all_reduce"):
        monkeypatch.setattr(dist, name, getattr(group, name))
    results, errors = [None, None], []

    def rank_main(r):
        group.turn.acquire()
        try:
            group.local.rank = r
            torch.cuda.set_device(0)
            assert dp.exchange_active() and dp.world_size() == 2 and dp.block_shard(32) == (16 * r, 16 * r + 16)
            torch.manual_seed(0)
            m, _, _ = _model()
            rec = Recorder(m)
            loop = trainer.WatermarkLoop(m, rec, KW)
            out = loop.step(per_rank[r], msg)
            torch.cuda.synchronize()
            results[r] = (rec, [float(v.detach()) for v in out[3:6]], int(m.step_counter[0, 0]), loop.exchange.bytes_per_step, loop.exchange.collectives_per_step)
        except BaseException as e:      # noqa: BLE001 -- a dead rank must not leave the other one waiting at the barrier
            errors.append(e)
            group.barrier.abort()
        finally:
            if group.turn.locked():
                try:
                    group.turn.release()
                except RuntimeError:
                    pass

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not errors, errors
    (rec0, l0, n0, nbytes, ncoll), (rec1, l1, n1, _, _) = results
    assert rec0.n_selected == 32 and ncoll in (1, 2) and nbytes >= rec0.G.numel() * 4
    # ... decoded all 32 (same watermark loss everywhere, equal to the single-process one) ...
    np.testing.assert_allclose([l0[1], l1[1]], [ref_losses[1]] * 2, rtol=1e-5)
    np.testing.assert_allclose(0.5 * (l0[0] + l1[0]), ref_losses[0], rtol=1e-5)          # image loss: mean over the ranks
    # ... and after the sum all-reduce holds the single-process gradient (G: plain sum; decoder: sum / world, GradExchange(average=True))
    assert torch.equal(rec0.G, rec1.G)
    rel_G = float((rec0.G - ref.G).norm() / ref.G.norm())
    rel_dec = float((rec0.dec - ref.dec).norm() / ref.dec.norm())
    print(f"\ntwo-rank vs single-process: codebook gradient rel. L2 {rel_G:.2e}; decoder gradient rel. L2 {rel_dec:.2e}")
    assert rel_G <= 1e-5
    assert rel_dec <= 1e-4
    kinds = sorted({k for k, _, _ in group.log})
    assert kinds == ["all_gather", "all_reduce"]


def test_sharded_codebook_optimizer_equals_replicated(strict_mlp):
    """dp.optimizer_shard: rank r runs the fused codebook Adam on the tables of ITS bits only and contributes a partial pre-sum of the next
    message.  Emulated on one GPU with three replicas of the same tables and the same (all-reduced) gradient G: one replica takes the
    replicated step over all D bits, the other two play ranks 0 and 1.  Owned tables, their Adam moments and step counts must equal the
    replicated result bit for bit, the other rank's tables must stay untouched, and the two partial pre-sums must add up to the
    replicated one."""
    from nerf_signature_amd.optim import CodebookAdam
    D = 32
    torch.manual_seed(0)
    G = torch.randn(1 << 19, 2, device="cuda") * 1e-3
    msg = torch.from_numpy(cf.messages(D)[2]).cuda()
    nxt = (1 - msg).contiguous()
    nxt[::3] = msg[::3]
    lr = torch.tensor(1e-2, device="cuda")

    def replica():
        tables = [torch.nn.Parameter(torch.from_numpy(cf.table(100 + l, scale=0.05)).cuda()) for l in range(2 * D)]
        return tables, CodebookAdam([{"params": tables}], lr=1e-2, betas=(0.9, 0.99), eps=1e-15, capturable=True)

    t_ref, o_ref = replica()
    S_ref = torch.zeros(1 << 19, 2, device="cuda")
    for _ in range(2):                                  # two steps: the second one sees non-zero moments
        o_ref.step_shared_sel(t_ref, msg, G, lr, 1.0, next_message_dev=nxt, S_next=S_ref)
    parts = []
    for r in range(2):
        b0, b1 = r * D // 2, (r + 1) * D // 2
        t, o = replica()
        S = torch.zeros(1 << 19, 2, device="cuda")
        for _ in range(2):
            o.step_shared_sel(t[2 * b0:2 * b1], msg[b0:b1], G, lr, 1.0, next_message_dev=nxt[b0:b1], S_next=S)
        parts.append(S)
        init = [torch.from_numpy(cf.table(100 + l, scale=0.05)).cuda() for l in range(2 * D)]
        for l in range(2 * D):
            if 2 * b0 <= l < 2 * b1:
                assert torch.equal(t[l], t_ref[l])
                if len(o_ref.state[t_ref[l]]):
                    for k in ("exp_avg", "exp_avg_sq", "step"):
                        assert torch.equal(o.state[t[l]][k], o_ref.state[t_ref[l]][k])
            else:
                assert torch.equal(t[l], init[l]) and len(o.state[t[l]]) == 0
    np.testing.assert_allclose((parts[0] + parts[1]).cpu().numpy(), S_ref.cpu().numpy(), rtol=0, atol=1e-6)   # fp32 summation order (values up to ~1.6)
    assert float(S_ref.abs().max()) > 0.1


def test_stage1_loop_on_two_real_gloo_rank_processes():
    """tools/stage1_dp_check.py: two rank PROCESSES (gloo, sharing this GPU) drive the captured stage-1 loop -- packed gradient exchange between captured segments, device-side
    grid refresh, parameter EMA, and point buffers that only rank 0 fills to 95 %: both ranks grow at the same refresh; parameters, EMA shadows and the density grid stay
    identical on both ranks (checked inside the ranks with all_gather + torch.equal)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "stage1_dp_check.py")], env=env, capture_output=True, text=True, timeout=400)
    assert out.returncode == 0, out.stderr[-3000:]
    assert "identical on both ranks" in out.stdout and "on BOTH ranks at the same refresh (1 re-capture each)" in out.stdout, out.stdout[-1000:]
