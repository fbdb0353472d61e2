"""CPU tests of the data-parallel driver's host logic: batch split rule, LR/EMA schedules against the golden
values taken from the reference, and the N>1 path with world_size-2 gloo processes that run the product's
`DPTrainer.train_step` / `_exchange` / `GradExchange` (only the two device stages — HIP backward, fused optimizer kernel —
are replaced by CPU stand-ins)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.util import golden


def test_split_rule_and_schedules():
    from bsi_amd import dp
    # bsi/data/h5image.py:312
    assert [dp.split_batch(512, 8, r) for r in range(8)] == [64] * 8
    assert [dp.split_batch(10, 4, r) for r in range(4)] == [3, 3, 2, 2]
    assert sum(dp.split_batch(129, 8, r) for r in range(8)) == 129
    # bsi/data/sampler.py:63
    assert dp.rank_indices(10, 4, 1) == [1, 5, 9]
    g = golden("g8_optimizer")
    lrs = g["lr_schedule"].tolist()
    for step, ref in enumerate(lrs):
        got = dp.warmup_cosine_lr(step, base_lr=5e-4, warmup_steps=10, max_steps=60, start_lr=1e-8, end_lr=5e-5)
        assert abs(got - ref) <= 1e-9 + 1e-6 * abs(ref), (step, got, ref)
    # EMA: decay actually applied by the reference at each update() call (ema_steps = self.step before the call)
    for s, d in zip(g["ema_steps"].tolist(), g["ema_decays"].tolist()):
        w = dp.ema_weight(int(s))
        if s <= 1000:
            assert w == 1.0          # copy_params_from_model_to_ema
        else:
            assert abs((1.0 - w) - d) < 1e-12, (s, w, d)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class _ToyDiT(torch.nn.Module):
    """Parameter layout of the DiT as the trainer sees it: dit.patch_encoder, dit.blocks.{i}, dit.patch_decoder."""

    def __init__(self, depth=3):
        super().__init__()
        self.dit = torch.nn.Module()
        self.dit.patch_encoder = torch.nn.Linear(6, 8)
        self.dit.blocks = torch.nn.ModuleList([torch.nn.Linear(8, 8) for _ in range(depth)])
        self.dit.patch_decoder = torch.nn.Linear(8, 6)

    def forward(self, x):
        h = self.dit.patch_encoder(x)
        for b in self.dit.blocks:
            h = h + torch.tanh(b(h))
        return self.dit.patch_decoder(h)


class _ToyBSI:
    def __init__(self, model):
        self.model = model

    def train_loss(self, x, generator=None):
        return (self.model(x) - x).square().mean(dim=1)


def _host_trainer(bsi, log, **kw):
    """The product's DPTrainer with its two DEVICE stages replaced by CPU stand-ins (autograd for the HIP backward, the
    oracle's clip+AdamW restatement for the fused kernel).  `train_step`, `_exchange`, the bucket plan, the gate order and
    the 1/world scale are the product's own code."""
    from bsi_amd import dp
    from oracle.bsi_oracle import clip_adamw_step

    class HostTrainer(dp.DPTrainer):
        def _setup_update_state(self, dev):
            self.m, self.v = torch.zeros_like(self.fp.flat), torch.zeros_like(self.fp.flat)

        def _setup_exchange_state(self, dev, depth):
            self.comm_stream, self.events = None, None

        def _backward(self, x, generator):
            for p in self.model.parameters():
                p.grad = None
            loss = self.bsi.train_loss(x, generator).mean()
            loss.backward()
            return loss, torch.cat([p.grad.reshape(-1) for p in self.model.parameters()])

        def _gate_wait(self, l):
            log.append(("gate", l))

        def _update(self, flat_g, lr, ema_w):
            g = flat_g * (1.0 / self.world)
            P, G, M, V = [self.fp.flat], [g], [self.m], [self.v]
            clip_adamw_step(P, G, M, V, self.step_count, lr=lr, beta1=self.betas[0], beta2=self.betas[1], eps=self.eps,
                            weight_decay=self.weight_decay, max_norm=self.max_grad_norm)
            if self.ema_fp is not None and ema_w >= 0:
                self.ema_fp.flat.lerp_(self.fp.flat, ema_w)

    return HostTrainer(bsi, **kw)


def _data(n):
    return torch.randn(n, 6, generator=torch.Generator().manual_seed(3))


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from bsi_amd import dp
    torch.manual_seed(0)                     # identical initial weights on every rank
    model = _ToyDiT()
    log = []
    real_all_reduce = dist.all_reduce

    def spy(t, *a, **kw):
        log.append(("allreduce", t.numel()))
        return real_all_reduce(t, *a, **kw)

    dist.all_reduce = spy
    tr = _host_trainer(_ToyBSI(model), log, lr=1e-2, betas=(0.9, 0.99), weight_decay=1e-2, max_grad_norm=0.05)
    assert tr.world == world and tr.exchange and tr.bucketed
    X = _data(129)                           # uneven split 129 -> 65 / 64 (bsi/data/h5image.py:309-312)
    nb = dp.split_batch(129, world, rank)
    start = sum(dp.split_batch(129, world, r) for r in range(rank))
    losses = [float(tr.train_step(X[start:start + nb])) for _ in range(2)]
    dist.all_reduce = real_all_reduce
    ret[rank] = {"nb": nb, "flat": tr.fp.flat.clone(), "ema": tr.ema_fp.flat.clone(), "log": log, "losses": losses,
                 "plan": tr.xchg.plan, "n": tr.fp.flat.numel(), "steps": tr.step_count}
    dist.destroy_process_group()


def test_gloo_world2_dptrainer_step():
    """world_size 2 over gloo: the product's DPTrainer.train_step / _exchange / GradExchange on an uneven split."""
    from bsi_amd import dp
    from oracle.bsi_oracle import clip_adamw_step
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    a, b = ret[0], ret[1]
    assert (a["nb"], b["nb"]) == (65, 64) and a["steps"] == 2
    # identical parameters and EMA on both ranks after two steps
    assert torch.equal(a["flat"], b["flat"]) and torch.equal(a["ema"], b["ema"])
    assert a["losses"] != b["losses"]          # different shards
    # bucket order: blocks last-to-first, each gated on its own event, then encoder and decoder; the plan tiles the buffer
    depth = 3
    per_step = len(a["log"]) // 2
    step_log = a["log"][:per_step]
    assert [e for e in step_log if e[0] == "gate"] == [("gate", l) for l in reversed(range(depth))]
    kinds = [e[0] for e in step_log]
    assert kinds == ["gate", "allreduce"] * depth + ["allreduce"] * 2
    assert sum(e[1] for e in step_log if e[0] == "allreduce") == a["n"]
    assert [g for _, _, g in a["plan"]] == [2, 1, 0, None, None]
    # single-process restatement: DDP average of the per-rank MEAN-loss gradients (bsi/tasks/bsi.py:163-166), clip on the
    # averaged gradient, AdamW; EMA copies during warm-up
    torch.manual_seed(0)
    model = _ToyDiT()
    bsi = _ToyBSI(model)
    X = _data(129)
    P = [torch.cat([p.detach().reshape(-1) for p in model.parameters()])]
    M, V = [torch.zeros_like(P[0])], [torch.zeros_like(P[0])]
    for step in (1, 2):
        off = 0
        for p in model.parameters():     # load the current flat parameters
            p.data.copy_(P[0][off:off + p.numel()].view_as(p))
            off += p.numel()
        gsum = torch.zeros_like(P[0])
        for lo, hi in ((0, 65), (65, 129)):
            for p in model.parameters():
                p.grad = None
            bsi.train_loss(X[lo:hi]).mean().backward()
            gsum += torch.cat([p.grad.reshape(-1) for p in model.parameters()])
        clip_adamw_step(P, [gsum / world], M, V, step, lr=1e-2, beta1=0.9, beta2=0.99, eps=1e-8, weight_decay=1e-2,
                        max_norm=0.05)
    assert torch.allclose(a["flat"], P[0], rtol=1e-5, atol=1e-7)
    assert torch.equal(a["ema"], a["flat"])   # first 1000 updates copy the online weights (ema_pytorch.py:320-332)


def _worker_single_bucket(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(6, 8), torch.nn.Tanh(), torch.nn.Linear(8, 6))   # no `.dit`: one bucket
    log = []
    tr = _host_trainer(_ToyBSI(model), log, lr=1e-2, max_grad_norm=None, ema=False)
    assert not tr.bucketed and tr.xchg.plan == [(0, tr.fp.flat.numel(), None)]
    X = _data(10)
    tr.train_step(X[rank::world])            # DistributedNonPaddingSampler order (bsi/data/sampler.py:63)
    ret[rank] = tr.fp.flat.clone()
    dist.destroy_process_group()


def test_gloo_world2_single_bucket_model():
    """Models without per-block events (the UNet path): one bucket over the whole gradient."""
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_single_bucket, args=(2, _free_port(), ret), nprocs=2, join=True)
    assert torch.equal(ret[0], ret[1])
