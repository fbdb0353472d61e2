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


def test_summary_is_compact_and_tolerates_missing_parts():
    """bench.py's trailing `summary` key (the driver keeps the last 2000 characters of the line): built from whatever the line holds --
    a failed train benchmark, a skipped secondary block, a multi-GPU record -- and short enough to survive with the keys behind it."""
    import json
    bench = _bench()
    full = {"value": 48.9, "model_frac_of_peak": 0.41, "n_gpus": 1, "roofline": {"frac": 0.4578},
            "train": {"value": 3.76, "ms_per_step": 266.2, "model_tflops_per_gpu": 931.0, "optimizer_ms": 3.3,
                      "per_rank_workloads_on_one_gpu": [{"world": 8, "fwd_bwd_ms": 38.7, "fwd_bwd_ms_cu_reserve": 41.6, "optimizer_ms_sharded": 0.48}]},
            "secondary": {"vdm_unet": {"sample": {"value": 111.9}, "train": {"value": 25.9}}, "dit_l2_sample_128": {"value": 47.9}},
            "cpu_baseline": {"value": 0.0242}}
    s = bench.summary(full)
    assert s["train_steps_per_s"] == 3.76 and s["per_rank_ms"] == [[8, 38.7, 41.6, 0.48]] and s["unet_images_per_s"] == 111.9
    assert s["sample_256_128_64"] == [None, 47.9, None] and s["fc1_roofline_frac"] == 0.458 and abs(s["train_frac_of_peak"] - 0.37) < 0.01
    multi = {"value": 390.0, "n_gpus": 8, "roofline": {"frac": 0.45},
             "train": {"value": 20.0, "ms_per_step": 50.0, "model_tflops_per_gpu": 600.0, "optimizer_ms": 3.3,
                       "variants": {"allreduce": {"ms_per_step": 50.0}, "sharded_update": {"error": "x"}},
                       "comm": {"exposed_comm_ms": 4.0, "bucket_allreduce_alone": {"busbw_GBps": 280.0}}}}
    m = bench.summary(multi)
    assert m["train_variants_ms"] == {"allreduce": 50.0, "sharded_update": None} and m["exposed_comm_ms"] == 4.0 and m["bucket_busbw_GBps"] == 280.0
    broken = bench.summary({"value": 48.0, "train": {"error": "RuntimeError: boom"}, "secondary": {"error": "child died"}})
    assert broken["train_error"] == "RuntimeError: boom" and broken["train_steps_per_s"] is None and broken["unet_images_per_s"] is None
    assert max(len(json.dumps(x)) for x in (s, m, broken)) < 1200
