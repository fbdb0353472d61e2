#!/usr/bin/env python
"""Timing of the weight-gradient GEMM dW = dY^T X (+ bias gradient) on the DiT-L shapes at the training batch
(B images x 256 tokens).  One mode per process (BSI_TN_ABL is read once); A/B = run it twice on one box."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bsi_amd import _native as N  # noqa: E402

dev = "cuda"
B = int(os.environ.get("B", "512"))
M = B * 256
SHAPES = [("qkv", 3072, 1024), ("out", 1024, 1024), ("fc1", 4096, 1024), ("fc2", 1024, 4096)]
ROUNDS, ITERS = 5, 5
lib = N.lib()
g = torch.Generator(device=dev).manual_seed(0)
for name, Nn, K in SHAPES:
    dY = torch.randn((M, Nn), device=dev, generator=g).to(torch.bfloat16)
    X = torch.randn((M, K), device=dev, generator=g).to(torch.bfloat16)
    out = torch.empty((Nn, K), device=dev)
    cs = torch.empty(Nn, device=dev)
    ws = torch.empty(lib.bsi_gemm_tn_workspace_bytes(M, Nn, K), dtype=torch.uint8, device=dev)
    def run():
        if os.environ.get("NOBIAS"):
            N.check(lib.bsi_gemm_tn_bf16(N.ptr(dY), Nn, N.ptr(X), K, M, Nn, K, N.ptr(out), K, 0, N.ptr(ws), N.stream()))
        else:
            N.check(lib.bsi_gemm_tn_bias_bf16(N.ptr(dY), Nn, N.ptr(X), K, M, Nn, K, N.ptr(out), K, N.ptr(cs), 0, N.ptr(ws), N.stream()))
    run(); run()
    torch.cuda.synchronize()
    ref = dY[:4096].float().t() @ X[:4096].float()
    ms = []
    for r in range(ROUNDS):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(ITERS):
            run()
        e1.record()
        torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1) / ITERS)
    ms.sort()
    flops = 2.0 * M * Nn * K
    print(f"{name:4s} M={M} N={Nn} K={K}: med {flops / ms[len(ms) // 2] / 1e9:7.0f} TF (best {flops / ms[0] / 1e9:6.0f})  {ms[len(ms) // 2] * 1e3:8.1f} us"
          f"  checksum {float(out.double().abs().sum()):.6e}  ABL={os.environ.get('BSI_TN_ABL', '0')} NOBIAS={os.environ.get('NOBIAS', '')}", flush=True)

# the paired launch of a block's qkv and out-projection gradients (round 5) against the two single launches
dY1 = torch.randn((M, 3072), device=dev, generator=g).to(torch.bfloat16)
dY2 = torch.randn((M, 1024), device=dev, generator=g).to(torch.bfloat16)
X1 = torch.randn((M, 1024), device=dev, generator=g).to(torch.bfloat16)
X2 = torch.randn((M, 1024), device=dev, generator=g).to(torch.bfloat16)
o1, o2 = torch.empty((3072, 1024), device=dev), torch.empty((1024, 1024), device=dev)
ws = torch.empty(lib.bsi_gemm_tn_workspace_bytes(M, 4096, 1024), dtype=torch.uint8, device=dev)


def pair():
    N.check(lib.bsi_gemm_tn_pair_bf16(N.ptr(dY1), 3072, N.ptr(X1), 3072, N.ptr(o1), N.ptr(dY2), 1024, N.ptr(X2), 1024, N.ptr(o2), 1024, M, 1024,
                                      N.ptr(ws), N.stream()))


def two():
    N.check(lib.bsi_gemm_tn_bf16(N.ptr(dY1), 3072, N.ptr(X1), 1024, M, 3072, 1024, N.ptr(o1), 1024, 0, N.ptr(ws), N.stream()))
    N.check(lib.bsi_gemm_tn_bf16(N.ptr(dY2), 1024, N.ptr(X2), 1024, M, 1024, 1024, N.ptr(o2), 1024, 0, N.ptr(ws), N.stream()))


res = {"pair": [], "two launches": []}
for r in range(ROUNDS):
    for name, fn in (("pair", pair), ("two launches", two)):
        fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(ITERS):
            fn()
        e1.record()
        torch.cuda.synchronize()
        res[name].append(e0.elapsed_time(e1) / ITERS)
flops = 2.0 * M * 4096 * 1024
for name, ms in res.items():
    ms.sort()
    print(f"qkv + out ({name}) M={M}: med {flops / ms[len(ms) // 2] / 1e9:7.0f} TF  {ms[len(ms) // 2] * 1e3:8.1f} us (slab sums included)", flush=True)
