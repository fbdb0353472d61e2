"""World size 2 with the REAL device stages on one GPU: two processes on cuda:0 in a gloo group over device tensors run the
unmodified DPTrainer.train_step (tests/dp_gloo_worker.py).  Checked: parameters and EMA bit-identical on both ranks, and equal
to a single-process restatement of DistributedDataParallel's semantics on the whole batch (/root/reference/bsi/tasks/bsi.py:
163-166: the average over ranks of each rank's mean-loss gradient, then ONE clip + AdamW + EMA update, :187-198)."""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_processes_on_one_gpu_run_the_product_train_step(tmp_path):
    from tests.util import report
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    out = str(tmp_path / "dp")
    procs = []
    for r in range(2):  # fresh processes; they make their first GPU call themselves
        env = dict(os.environ, RANK=str(r), LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   DP_OUT=out)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dp_gloo_worker.py")], env=env, cwd=ROOT,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = []
    try:
        for p in procs:
            o, _ = p.communicate(timeout=600)
            logs.append(o.decode()[-3000:])
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    assert [p.returncode for p in procs] == [0, 0], logs
    a0, a1 = (torch.load(out + f".rank{r}.pt") for r in range(2))
    for mode in ("allreduce", "sharded", "sharded_overlap"):
        assert torch.equal(a0[mode]["flat"], a1[mode]["flat"]) and torch.equal(a0[mode]["ema"], a1[mode]["ema"]), mode  # replicas stay bit-identical
    # the sharded step (reduce-scatter -> bsi_clip_adamw_ema_segments on the rank's slices -> all-gather) == the all-reduce step, bit for bit
    assert torch.equal(a0["sharded"]["flat"], a0["allreduce"]["flat"]) and torch.equal(a0["sharded"]["ema"], a0["allreduce"]["ema"])
    assert a0["sharded"]["sq"] == a0["allreduce"]["sq"] and a0["sharded"]["losses"] == a0["allreduce"]["losses"]
    # ... and so is the sharded step whose all-gather overlaps the next (gated) forward
    for k in ("flat", "ema", "sq", "losses"):
        same = torch.equal(a0["sharded_overlap"][k], a0["sharded"][k]) if torch.is_tensor(a0["sharded"][k]) else a0["sharded_overlap"][k] == a0["sharded"][k]
        assert same, f"sharded_overlap differs from sharded in {k}"
    r0, r1 = a0["allreduce"], a1["allreduce"]
    assert r0["buckets"] == 2 + 2  # one per block (last first) + patch encoder + decoder

    # single-process restatement: both shards' gradients with the same draws, summed, 1/2 in the fused update
    from bsi_amd.dp import DPTrainer, split_batch
    from tests.dp_gloo_worker import STEPS, TRAINER, build
    from tests.util import golden, replay_draws, shard_draws
    dev = torch.device("cuda", 0)
    g = golden("g4_train_dit")
    tr = DPTrainer(build(dev), rehearse=(2, 0), **TRAINER)  # buckets / slices / norm chunks laid out as at world 2; the update
    assert tr.world == 1 and not tr.exchange                # divides the SUM of the two shard gradients by 2 (DDP's average)
    B = g["x"].shape[0]
    from bsi_amd.dp import ema_weight
    losses = [[], []]
    for s in range(STEPS):
        total = None
        for rank in range(2):
            nb = split_batch(B, 2, rank)
            start = sum(split_batch(B, 2, r) for r in range(rank))
            off, perm, eps = shard_draws(rank, s, nb, (3, 16, 16))
            with replay_draws(dev, rand=[off], randperm=[perm], randn=[eps]):
                loss, flat_g = tr._backward(g["x"][start:start + nb].to(dev), None)
            losses[rank].append(float(loss))
            total = flat_g.clone() if total is None else total + flat_g
        w = ema_weight(tr.step_count, beta=tr.ema_beta, update_after_step=tr.ema_after)
        tr.step_count += 1
        tr._update(total, tr.lr, w)
        tr._invalidate(tr.model)
        tr._invalidate(tr.ema_model)
    torch.cuda.synchronize()
    dp, de = (float((a - b).abs().max() / b.abs().max()) for a, b in ((r0["flat"], tr.fp.flat.cpu()), (r0["ema"], tr.ema_fp.flat.cpu())))
    report("dp_world2_one_gpu", params_rel_linf_vs_single_process=dp, ema_rel_linf_vs_single_process=de,
           rank_losses=[r0["losses"], r1["losses"]], single_process_losses=losses, buckets=r0["buckets"])
    assert r0["losses"] == losses[0] and r1["losses"] == losses[1]
    assert dp <= 1e-6 and de <= 1e-6, (dp, de)
