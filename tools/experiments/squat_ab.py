#!/usr/bin/env python
"""One-GPU rehearsal of the data-parallel step's CU sharing.

`DPTrainer.train_step` (DiT-L/2) runs on the main stream while a "squatter" kernel (tools/experiments/squatter.hip: one
160-KB-LDS workgroup per CU, spinning) holds S compute units on a HIGH-PRIORITY side stream for the length of the step --
what the RCCL all-reduce kernels of `_exchange` do on a multi-GPU node.  Arms, interleaved in ONE process, median of the
timed steps:

    free          no squatter, no reserve                 (the single-GPU step)
    reserve R     no squatter, bsi_set_cu_reserve(R)      (what the reserve costs by itself)
    squat S       squatter on S CUs, no reserve           (static tile partition meets taken CUs)
    squat S + R   squatter on S CUs, reserve R = S        (the product's data-parallel configuration)

Done when `squat S + reserve S` is within S/256 + 3 % of `free`.  Usage (GPU box):
    python tools/experiments/squat_ab.py [B=256] > profiles/r4/cu_reserve_squatter_ab.txt
"""
import ctypes as C
import os
import statistics
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from bsi_amd import BSI, Discretization  # noqa: E402
from bsi_amd import _native as N  # noqa: E402
from bsi_amd.dp import DPTrainer  # noqa: E402


def squatter_lib():
    src = os.path.join(ROOT, "tools", "experiments", "squatter.hip")
    out = os.path.join(ROOT, "tools", "experiments", "build", "libsquatter.so")
    if not os.path.isfile(out) or os.path.getmtime(out) < os.path.getmtime(src):
        os.makedirs(os.path.dirname(out), exist_ok=True)
        subprocess.run(["/opt/rocm/bin/hipcc", "-O2", "-shared", "-fPIC", "--offload-arch=gfx950", src, "-o", out], check=True)
    lib = C.CDLL(out)
    lib.squat_launch.restype = C.c_int
    lib.squat_launch.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p]
    return lib


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    steps = int(os.environ.get("STEPS", "5"))
    rounds = int(os.environ.get("ROUNDS", "3"))
    piece_us = int(os.environ.get("PIECE_US", "2000"))     # one squatter launch ~ one bucket of the all-reduce
    lds = int(os.environ.get("SQUAT_LDS", str(160 * 1024)))
    dev = torch.device("cuda", 0)
    sq = squatter_lib()
    model, shape = bench.build_model(dev)
    bsi = BSI(model, data_shape=shape, lambda_0=1e-2, alpha_M=1e6, alpha_R=2e6, k=128, preconditioning="edm",
              discretization=Discretization.image_8bit()).to(dev)
    model.train()
    tr = DPTrainer(bsi, lr=5e-4, betas=(0.9, 0.99), weight_decay=1e-2, max_grad_norm=1.0, cu_reserve=0, tile_queue=False)
    g = torch.Generator(dev).manual_seed(0)
    x = (torch.round(255 * torch.rand((B, *shape), device=dev, generator=g)) / 255) * 2 - 1
    side = torch.cuda.Stream(device=dev, priority=-1)      # high priority, like a communication stream
    for _ in range(2):
        tr.train_step(x, g)
    torch.cuda.synchronize()

    def one_step(squat, reserve, est_ms, queue=0, duty=1.0):
        tr.cu_reserve = reserve
        tr.tile_queue = bool(queue)
        n_pieces = 0
        gap_us = int(piece_us * (1.0 / duty - 1.0)) if squat else 0   # duty < 1: the squatter holds its CUs `duty` of the time
        if squat:
            # enough back-to-back pieces to cover the whole step (they queue on the side stream and run one after the other)
            n_pieces = int(est_ms * 1.6 * 1000 / (piece_us + gap_us)) + 2
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n_pieces):
            assert sq.squat_launch(squat, piece_us, lds, C.c_void_p(side.cuda_stream)) == 0
            if gap_us:  # the gap: a one-workgroup kernel without LDS (an idle communication stream would hold nothing at all)
                assert sq.squat_launch(1, gap_us, 0, C.c_void_p(side.cuda_stream)) == 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        tr.train_step(x, g)
        e1.record()
        e1.synchronize()
        ms = e0.elapsed_time(e1)
        torch.cuda.synchronize()                            # the rest of the squatter pieces drain here, outside the measurement
        return ms, 1e3 * (time.perf_counter() - t0)

    # (name, squatter CUs, reserve, tile queue, squatter duty)
    arms = [("free", 0, 0, 0, 1.0), ("free, queue", 0, 0, 1, 1.0), ("reserve 16", 0, 16, 0, 1.0), ("reserve 16, queue", 0, 16, 1, 1.0),
            ("squat 16", 16, 0, 0, 1.0), ("squat 16, queue", 16, 0, 1, 1.0), ("squat 16 + reserve 16", 16, 16, 0, 1.0),
            ("squat 16 + reserve 16, queue", 16, 16, 1, 1.0),
            ("squat 16 at 30 % duty + reserve 16", 16, 16, 0, 0.3), ("squat 16 at 30 % duty + reserve 16, queue", 16, 16, 1, 0.3),
            ("squat 16 at 30 % duty, queue", 16, 0, 1, 0.3)]
    if os.environ.get("ARMS"):
        keep = os.environ["ARMS"].split(";")
        arms = [a for a in arms if a[0] in keep or a[0] == "free"]
    res = {a[0]: [] for a in arms}
    base = one_step(0, 0, 0)[0]
    for _ in range(rounds):
        for name, s_, r_, q_, d_ in arms:
            for _ in range(steps):
                res[name].append(one_step(s_, r_, base * 2.2, q_, d_)[0])
    tr.tile_queue = False
    free = statistics.median(res["free"])
    print(f"# DiT-L/2 DPTrainer.train_step, per-GPU batch {B}, one MI355X ({N.lib().bsi_compute_cus()} CUs), squatter pieces of {piece_us} us "
          f"with {lds // 1024} KB of LDS per workgroup on a high-priority stream; {rounds} interleaved rounds x {steps} steps, ms per step")
    print(f"{'arm':44s} {'median':>9s} {'min':>9s} {'max':>9s} {'vs free':>9s}   allowed (CUs taken / 256 + 3 %)")
    for name, s_, r_, q_, d_ in arms:
        v = res[name]
        med = statistics.median(v)
        allowed = f"{100 * (max(s_, r_) / 256 + 0.03):5.1f} %" if (s_ or r_) else ""
        print(f"{name:44s} {med:9.2f} {min(v):9.2f} {max(v):9.2f} {100 * (med / free - 1):+8.1f} %   {allowed}")


if __name__ == "__main__":
    main()
