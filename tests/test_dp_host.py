"""CPU tests of the data-parallel driver's host logic: batch split rule, LR/EMA schedules against the golden
values taken from the reference, and the N>1 gradient exchange with world_size-2 gloo processes."""
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


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from bsi_amd import dp
    torch.manual_seed(0)
    # a toy "model": loss = mean over the shard of (w . x)^2; DDP semantics = average of per-rank mean-loss gradients
    w = torch.arange(1, 7, dtype=torch.float32) / 10
    X = torch.randn(10, 6, generator=torch.Generator().manual_seed(1))
    nb = dp.split_batch(10, world, rank)
    start = sum(dp.split_batch(10, world, r) for r in range(rank))
    xs = X[start:start + nb]
    wr = w.clone().requires_grad_(True)
    ((xs @ wr) ** 2).mean().backward()
    flat = torch.cat([wr.grad, torch.full((3,), float(rank + 1))])
    buckets = [flat[:4], flat[4:]]           # views of one flat buffer, as the trainer's block spans
    dp.allreduce_sum_buckets(buckets)
    flat /= world
    ret[rank] = flat.clone()
    dist.destroy_process_group()


def test_gloo_world2_gradient_average():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    a, b = ret[0], ret[1]
    assert torch.equal(a, b)                       # every rank holds the same averaged gradient
    w = (torch.arange(1, 7, dtype=torch.float32) / 10).requires_grad_(True)
    X = torch.randn(10, 6, generator=torch.Generator().manual_seed(1))
    ((X @ w) ** 2).mean().backward()               # equal shards -> average of shard means == global mean
    assert torch.allclose(a[:6], w.grad, rtol=1e-6, atol=1e-7)
    assert torch.allclose(a[6:], torch.full((3,), 1.5))
