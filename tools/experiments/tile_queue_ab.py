#!/usr/bin/env python
"""A/B of the persistent GEMM's tile queue (bsi_set_tile_queue) against its static tile shares, per DiT-L/2 GEMM shape, alone and next
to a squatter that holds S CUs on a high-priority stream (tools/experiments/squatter.hip -- what an RCCL kernel does while a gradient
bucket is in flight).  Interleaved rounds in one process; us per launch (median).  Usage (GPU box):
    B=64 S=16 python tools/experiments/tile_queue_ab.py"""
import ctypes as C
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools", "experiments"))
from bsi_amd import _native as N  # noqa: E402
from squat_ab import squatter_lib  # noqa: E402

dev = "cuda"
B = int(os.environ.get("B", "64"))
S = int(os.environ.get("S", "16"))
M = B * 256
SHAPES = [("qkv", 3072, 1024, N.EPI_BIAS_BF16), ("out", 1024, 1024, N.EPI_BIAS_BF16), ("fc1", 4096, 1024, N.EPI_BIAS_GELU_BF16),
          ("fc2", 1024, 4096, N.EPI_BIAS_BF16), ("dqkv", 1024, 3072, N.EPI_BIAS_BF16)]
ROUNDS, ITERS = 5, 20
lib = N.lib()
sq = squatter_lib()
side = torch.cuda.Stream(priority=-1)
g = torch.Generator(device=dev).manual_seed(0)
print(f"# M = {M} ({B} images), {lib.bsi_compute_cus()} CUs, squatter on {S} CUs; us per launch, median of {ROUNDS} rounds x {ITERS} launches")
print(f"{'gemm':6s} {'tiles':>6s} | {'static':>8s} {'queue':>8s} | {'static+reserve':>14s} | {'squat: static':>13s} {'static+reserve':>14s} {'queue':>8s}")
for name, Nn, K, epi in SHAPES:
    A = torch.randn((M, K), device=dev, generator=g).to(torch.bfloat16)
    W = (torch.randn((Nn, K), device=dev, generator=g) / K ** 0.5).to(torch.bfloat16)
    bias = torch.randn(Nn, device=dev, generator=g)
    out = torch.empty((M, Nn), device=dev, dtype=torch.bfloat16)
    args = N.GemmArgs(A=A.data_ptr(), W=W.data_ptr(), bias=bias.data_ptr(), out=out.data_ptr(), M=M, N=Nn, K=K, lda=K, ldw=K, ldo=Nn, epilogue=epi)
    arms = [("static", 0, 0, 0), ("queue", 1, 0, 0), ("static+reserve", 0, S, 0), ("squat static", 0, 0, S), ("squat static+reserve", 0, S, S),
            ("squat queue", 1, 0, S)]
    res = {a[0]: [] for a in arms}
    ref = None
    for r in range(ROUNDS):
        for an, q, rsv, squat in arms:
            N.check(lib.bsi_set_tile_queue(q))
            N.check(lib.bsi_set_cu_reserve(rsv))
            torch.cuda.synchronize()
            if squat:   # one piece long enough to cover the timed launches
                assert sq.squat_launch(squat, 20000, 160 * 1024, C.c_void_p(side.cuda_stream)) == 0
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            N.check(lib.bsi_gemm_bf16(C.byref(args), N.stream()))
            e0.record()
            for _ in range(ITERS):
                N.check(lib.bsi_gemm_bf16(C.byref(args), N.stream()))
            e1.record()
            e1.synchronize()
            res[an].append(1e3 * e0.elapsed_time(e1) / ITERS)
            torch.cuda.synchronize()
            chk = out.float().sum().item()
            ref = chk if ref is None else ref
            assert chk == ref, (an, chk, ref)   # every arm computes the same bits
    N.check(lib.bsi_set_tile_queue(0))
    N.check(lib.bsi_set_cu_reserve(0))
    med = {k: statistics.median(v) for k, v in res.items()}
    tiles = ((M + 255) // 256) * ((Nn + 255) // 256)
    print(f"{name:6s} {tiles:6d} | {med['static']:8.1f} {med['queue']:8.1f} | {med['static+reserve']:14.1f} | {med['squat static']:13.1f} "
          f"{med['squat static+reserve']:14.1f} {med['squat queue']:8.1f}", flush=True)
