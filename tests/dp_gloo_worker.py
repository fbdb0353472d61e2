"""Rank process of tests/test_hip_dp_one_gpu.py: TWO processes on ONE GPU (cuda:0) in a gloo group over device tensors, each
running the UNMODIFIED DPTrainer.train_step of the product on its shard of the g4 golden batch -- the HIP backward with
per-block events (bsi_dit_backward_set_events), the bucket plan on the side stream, the fused clip + AdamW + EMA kernel with
1/world -- i.e. the device stages that the CPU gloo tests of tests/test_dp_host.py replace by stand-ins, at world size 2 on
the hardware that is there.  Replaces DistributedDataParallel + optimizer + EMA of /root/reference/bsi/tasks/bsi.py:163-198.
Each rank writes its parameters / EMA / losses to $DP_OUT.rank<r>.pt."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def build(dev):
    from bsi_amd import BSI, Discretization
    from bsi_amd.models.dit import DenoisingDiT
    from bsi_amd.nn import FourierFeatures
    from tests.util import weights

    model = DenoisingDiT((3, 16, 16), 2, 128, 2, 2, dropout=None, fourier_features=FourierFeatures(n_min=6, n_max=8))
    model.load_state_dict(weights("dit_ff"))
    model = model.to(dev).train()
    bsi = BSI(model, data_shape=(3, 16, 16), lambda_0=1e-2, alpha_M=1e6, alpha_R=2e6, k=16, preconditioning="edm",
              discretization=Discretization.image_8bit()).to(dev)
    return bsi


TRAINER = dict(lr=5e-4, betas=(0.9, 0.99), weight_decay=1e-2, max_grad_norm=1.0, ema_update_after_step=0)
STEPS = 2


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dev = torch.device("cuda", 0)  # every rank on the one device
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo")
    from bsi_amd.dp import DPTrainer, split_batch
    from tests.util import golden, replay_draws, shard_draws

    g = golden("g4_train_dit")
    B = g["x"].shape[0]
    nb = split_batch(B, world, rank)
    start = sum(split_batch(B, world, r) for r in range(rank))
    x = g["x"][start:start + nb].to(dev)
    out = {}
    # the all-reduce step, then -- from the same start state and draws -- the sharded step (reduce-scatter, slice update, all-gather)
    # and the sharded step whose parameter all-gather is not waited for: the next forward waits block by block (gated forward)
    for mode, kw in (("allreduce", {}), ("sharded", {"shard_update": True}),
                     ("sharded_overlap", {"shard_update": True, "overlap_gather": True})):
        bsi = build(dev)
        tr = DPTrainer(bsi, **TRAINER, **kw)
        assert tr.world == world and tr.exchange and tr.bucketed and tr.comm_stream is not None and tr.events is not None
        losses = []
        for s in range(STEPS):
            off, perm, eps = shard_draws(rank, s, nb, (3, 16, 16))
            with replay_draws(dev, rand=[off], randperm=[perm], randn=[eps]):
                losses.append(float(tr.train_step(x)))
        tr.sync_params()
        tr.gather_ema()
        torch.cuda.synchronize()
        out[mode] = {"flat": tr.fp.flat.cpu(), "ema": tr.ema_fp.flat.cpu(), "losses": losses, "buckets": len(tr.xchg.plan),
                     "sq": float(tr.sq[0]), "n_params": tr.fp.n}
    torch.save(out, os.environ["DP_OUT"] + f".rank{rank}.pt")
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
