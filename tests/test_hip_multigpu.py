"""Multi-GPU tests over RCCL: skipped unless at least two devices are visible (the 1-GPU test box skips them; the
driver's multi-GPU tier and any 8-GPU node run them).  One process per GPU, started as children of the test process."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _need(n):
    if torch.cuda.device_count() < n:
        pytest.skip(f"needs {n} GPUs, {torch.cuda.device_count()} visible")


def test_rccl_two_rank_dptrainer_exchange():
    """DPTrainer on 2 ranks: exchanged gradient = sum over ranks (per-block buckets gated by the backward's events on the
    side stream), identical parameters / EMA on both ranks after two steps (bsi/tasks/bsi.py:163-198 semantics)."""
    _need(2)
    port = _port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dp_rccl_worker.py")], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, cwd=ROOT))
    out, _ = procs[0].communicate(timeout=600)
    codes = [p.wait(timeout=600) for p in procs]
    rec = json.loads(out.decode().strip().splitlines()[-1])
    assert codes == [0, 0] and rec["ok"], (codes, rec)


def test_bench_self_launch_two_ranks():
    """`python bench.py --gpus 2` (no torchrun): the parent starts both ranks itself and reports n_gpus = 2."""
    _need(2)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--batch", "8", "--k", "4", "--train-steps", "2", "--train-batch", "16"],
                       capture_output=True, timeout=1200, cwd=ROOT)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    line = json.loads(r.stdout.decode().strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["train"]["per_gpu_batch"] == 8


def test_bench_multi_rank_code_path_on_one_device():
    """The whole N > 1 path of bench.py (self-launch and supervision, barriers, max-over-ranks timing, DPTrainer exchange with
    per-block events on the side stream, the communication breakdown of the train record) with two ranks on the ONE GPU of the test
    box: BSI_BENCH_ONE_DEVICE=1 swaps RCCL for gloo over device tensors and puts both ranks on cuda:0.  Not a measurement."""
    env = dict(os.environ, BSI_BENCH_ONE_DEVICE="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--batch", "8", "--k", "4", "--train-steps", "2", "--train-batch", "16"],
                       capture_output=True, timeout=1200, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    line = json.loads(r.stdout.decode().strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and "test_hook" in line["config"]
    tr = line["train"]
    assert tr["per_gpu_batch"] == 8 and tr["comm"]["buckets"] == 24 + 2 and tr["comm"]["allreduce_bytes"] > 1.9e9
    assert tr["comm"]["ms_per_step_without_exchange"] > 0 and "exposed_comm_ms" in tr["comm"]
    # the four variants of the first hardware run all ran (a failed one carries {"error": ...}); the headline is the DDP-equivalent
    # all-reduce step, the quickest one is named beside it
    assert set(tr["variants"]) == {"allreduce", "allreduce_cu_reserve_16", "sharded_update", "sharded_update_overlap"}, tr["variants"]
    assert all("error" not in v and v["ms_per_step"] > 0 for v in tr["variants"].values()), tr["variants"]
    assert tr["headline_variant"] == "allreduce" and tr["ms_per_step"] == tr["variants"]["allreduce"]["ms_per_step"]
    assert tr["fastest_variant"] == min(tr["variants"], key=lambda n: tr["variants"][n]["ms_per_step"])
    assert tr["comm"]["bucket_allreduce_alone"] and line["summary"]["train_variants_ms"]
    assert "secondary" not in line and "cpu_baseline" not in line


def test_bench_under_the_drivers_launcher_on_one_device():
    """The driver's own N > 1 command line -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N --steps K --warmup W` -- with the one-device hook: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* come
    from the launcher's environment, bench.py must not start ranks of its own, and rank 0 alone prints the JSON line."""
    env = dict(os.environ, BSI_BENCH_ONE_DEVICE="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--batch", "8", "--k", "2", "--train-steps", "0", "--no-secondary", "--no-cpu-baseline"],
                       capture_output=True, timeout=1200, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, lines  # one JSON line for the whole job
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 1 and line["warmup"] == 0 and line["scaling"] == "weak"
    assert line["config"]["workload"] and line["value"] > 0
