"""bench.py's own launcher (`python bench.py --gpus N` without torchrun): a rank that dies must take the others down instead of
leaving them in a rendezvous until the collective timeout, and rank 0's stdout is what the caller sees.  No GPU, no torch
collective: the children are small scripts."""
import importlib.util
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.ONE_DEVICE = True  # the device-count refusal is covered below; the children here use no device
    return mod


def _child(tmp_path, body):
    path = tmp_path / "child.py"
    path.write_text("import os, sys, time\nrank = int(os.environ['RANK'])\n" + body)
    return [sys.executable, str(path)]


def test_failed_rank_terminates_the_others(tmp_path, capfd):
    bench = _bench()
    cmd = _child(tmp_path, "assert os.environ['MASTER_ADDR'] == '127.0.0.1' and os.environ['WORLD_SIZE'] == '3'\n"
                           "if rank == 1:\n    sys.exit(7)\nprint('rank0 waits', flush=True)\ntime.sleep(600)\n")
    t0 = time.monotonic()
    rc = bench.launch_ranks(types.SimpleNamespace(gpus=3, launch_timeout=120), command=cmd)
    assert rc == 7
    assert time.monotonic() - t0 < 60  # not the 600 s the surviving ranks would have slept
    assert "rank0 waits" in capfd.readouterr().out


def test_all_ranks_ok_relays_rank0_only(tmp_path, capfd):
    bench = _bench()
    cmd = _child(tmp_path, "print('line of rank', rank, flush=True)\n")
    assert bench.launch_ranks(types.SimpleNamespace(gpus=2, launch_timeout=120), command=cmd) == 0
    out = capfd.readouterr().out
    assert "line of rank 0" in out and "line of rank 1" not in out


def test_launch_timeout(tmp_path, capfd):
    bench = _bench()
    cmd = _child(tmp_path, "time.sleep(600)\n")
    t0 = time.monotonic()
    assert bench.launch_ranks(types.SimpleNamespace(gpus=2, launch_timeout=2), command=cmd) == 5
    assert time.monotonic() - t0 < 60


def test_refuses_more_ranks_than_devices(tmp_path, capfd):
    bench = _bench()
    bench.ONE_DEVICE = False
    import torch
    n = torch.cuda.device_count()
    assert bench.launch_ranks(types.SimpleNamespace(gpus=n + 1, launch_timeout=5), command=_child(tmp_path, "")) == 2
    assert "refusing" in capfd.readouterr().err
