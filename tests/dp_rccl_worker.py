"""Rank process of tests/test_hip_multigpu.py: one process per GPU over RCCL (env RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*).
Runs two DPTrainer steps of the small golden DiT on this rank's shard and checks, on the device:
  * the exchanged gradient equals the sum over ranks of the gradients each rank computed (all_gather of the pre-exchange
    copies), bucket by bucket, for the bucketed (DiT) path with per-block events;
  * parameters and EMA are bit-identical on all ranks after the steps.
Prints one JSON line on rank 0."""
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist.init_process_group("nccl", device_id=dev)
    from bsi_amd import BSI, Discretization
    from bsi_amd.dp import DPTrainer, split_batch
    from bsi_amd.models.dit import DenoisingDiT
    from bsi_amd.nn import FourierFeatures
    from tests.util import golden, weights

    g = golden("g4_train_dit")
    model = DenoisingDiT((3, 16, 16), 2, 128, 2, 2, dropout=None, fourier_features=FourierFeatures(n_min=6, n_max=8))
    model.load_state_dict(weights("dit_ff"))
    model = model.to(dev).train()
    bsi = BSI(model, data_shape=(3, 16, 16), lambda_0=1e-2, alpha_M=1e6, alpha_R=2e6, k=16, preconditioning="edm",
              discretization=Discretization.image_8bit()).to(dev)
    seen = {}

    class Probe(DPTrainer):
        def _exchange(self, flat_g):
            torch.cuda.current_stream().synchronize()
            seen["pre"] = flat_g.clone()
            super()._exchange(flat_g)
            torch.cuda.current_stream().synchronize()
            seen["post"] = flat_g.clone()

    force = world == 1  # a one-rank group (the worker run by hand on a one-GPU box): the exchange is forced on
    tr = Probe(bsi, lr=5e-4, betas=(0.9, 0.99), weight_decay=1e-2, max_grad_norm=1.0, force_exchange=force)
    assert tr.world == world and tr.exchange and tr.bucketed
    B = g["x"].shape[0]
    nb = split_batch(B, world, rank)
    start = sum(split_batch(B, world, r) for r in range(rank))
    x = g["x"][start:start + nb].to(dev)
    gen = torch.Generator(dev).manual_seed(100 + rank)
    ok = True
    worst = 0.0
    for _ in range(2):
        loss = tr.train_step(x, gen)
        pres = [torch.empty_like(seen["pre"]) for _ in range(world)]
        dist.all_gather(pres, seen["pre"])
        want = torch.stack(pres).sum(0)
        err = float((seen["post"] - want).abs().max() / want.abs().max())
        worst = max(worst, err)
        ok = ok and err < 1e-6 and bool(torch.isfinite(loss))
    flats = [torch.empty_like(tr.fp.flat) for _ in range(world)]
    emas = [torch.empty_like(tr.ema_fp.flat) for _ in range(world)]
    dist.all_gather(flats, tr.fp.flat)
    dist.all_gather(emas, tr.ema_fp.flat)
    same = all(torch.equal(flats[0], f) for f in flats) and all(torch.equal(emas[0], e) for e in emas)
    # the sharded step over RCCL (reduce-scatter, slice update, all-gather of the parameters) from the same start state and draws:
    # at world 2 a sum has one order, so parameters and EMA must equal the all-reduce step's bit for bit
    model2 = DenoisingDiT((3, 16, 16), 2, 128, 2, 2, dropout=None, fourier_features=FourierFeatures(n_min=6, n_max=8))
    model2.load_state_dict(weights("dit_ff"))
    bsi2 = BSI(model2.to(dev).train(), data_shape=(3, 16, 16), lambda_0=1e-2, alpha_M=1e6, alpha_R=2e6, k=16, preconditioning="edm",
               discretization=Discretization.image_8bit()).to(dev)
    tr2 = DPTrainer(bsi2, lr=5e-4, betas=(0.9, 0.99), weight_decay=1e-2, max_grad_norm=1.0, shard_update=True, force_exchange=force)
    gen = torch.Generator(dev).manual_seed(100 + rank)
    for _ in range(2):
        tr2.train_step(x, gen)
    tr2.gather_ema()
    sharded_same = bool(torch.equal(tr2.fp.flat, tr.fp.flat) and torch.equal(tr2.ema_fp.flat, tr.ema_fp.flat)) if world <= 2 else \
        bool(torch.allclose(tr2.fp.flat, tr.fp.flat, rtol=1e-5, atol=1e-7))
    same = same and sharded_same
    if rank == 0:
        print(json.dumps({"ok": bool(ok and same), "exchange_rel_err": worst, "identical_params": bool(same), "sharded_equals_allreduce": sharded_same,
                          "world": world, "per_rank_batch": nb}), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if ok and same else 1)


if __name__ == "__main__":
    main()
