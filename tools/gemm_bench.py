#!/usr/bin/env python
"""A/B timing of the bf16 GEMM variants on the DiT-L shapes (random operands), interleaved rounds in one
process (guide §5.4 rule 24).  torch.matmul (hipBLASLt) is timed beside them only as a same-chip yardstick;
it is not part of the product."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bsi_amd import _native as N  # noqa: E402

dev = "cuda"
B = int(os.environ.get("B", "256"))
M = B * 256
SHAPES = [("qkv", 3072, 1024, N.EPI_BIAS_BF16), ("out", 1024, 1024, N.EPI_BIAS_BF16),
          ("fc1", 4096, 1024, N.EPI_BIAS_GELU_BF16), ("fc2", 1024, 4096, N.EPI_BIAS_BF16)]  # the engine's epilogues
VARIANTS = [int(v) for v in os.environ.get("VARIANTS", "1,3").split(",")]
ROUNDS, ITERS = 5, 10
lib = N.lib()
g = torch.Generator(device=dev).manual_seed(0)

for name, Nn, K, epi in SHAPES:
    A = torch.randn((M, K), device=dev, generator=g).to(torch.bfloat16)
    W = (torch.randn((Nn, K), device=dev, generator=g) / K ** 0.5).to(torch.bfloat16)
    bias = torch.randn(Nn, device=dev, generator=g)
    gate = torch.randn((1, Nn), device=dev, generator=g) * 0.1
    out_bf = torch.empty((M, Nn), device=dev, dtype=torch.bfloat16)
    x = torch.zeros((M, Nn), device=dev)
    args = N.GemmArgs(A=A.data_ptr(), W=W.data_ptr(), bias=bias.data_ptr(), M=M, N=Nn, K=K, lda=K, ldw=K, ldo=Nn,
                      epilogue=epi, gate=gate.data_ptr(), gate_rows=1, gate_stride=Nn, tokens=256)
    args.out = x.data_ptr() if epi == N.EPI_GATE_RESID else out_bf.data_ptr()
    flops = 2.0 * M * Nn * K
    # correctness of every variant against fp32 torch on a slice
    ref = (A[:512].float() @ W.float().t() + bias)
    res = {}
    for v in VARIANTS:
        N.check(lib.bsi_gemm_set_variant(v))
        x.zero_()
        N.check(lib.bsi_gemm_bf16(C.byref(args), N.stream()))
        torch.cuda.synchronize()
        if epi == N.EPI_GATE_RESID:
            got = x[:512] / gate
        elif epi == N.EPI_BIAS_GELU_BF16:
            got, ref_c = out_bf[:512].float(), torch.nn.functional.gelu(ref, approximate="tanh")
        else:
            got = out_bf[:512].float()
        rc = ref_c if epi == N.EPI_BIAS_GELU_BF16 else ref
        err = float((got - rc).abs().max() / rc.abs().max())
        res[v] = {"err": err, "ms": []}
    tms = []
    for r in range(ROUNDS):
        for v in VARIANTS:
            N.check(lib.bsi_gemm_set_variant(v))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(ITERS):
                N.check(lib.bsi_gemm_bf16(C.byref(args), N.stream()))
            e1.record()
            torch.cuda.synchronize()
            res[v]["ms"].append(e0.elapsed_time(e1) / ITERS)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(ITERS):
            torch.matmul(A, W.t())
        e1.record()
        torch.cuda.synchronize()
        tms.append(e0.elapsed_time(e1) / ITERS)
    line = f"{name:4s} M={M} N={Nn} K={K}: "
    for v in VARIANTS:
        ms = sorted(res[v]["ms"])
        line += f"v{v & 255}s{(v >> 8) & 255}g{v >> 16} med {flops / ms[len(ms) // 2] / 1e9:7.0f} TF (best {flops / ms[0] / 1e9:6.0f}, err {res[v]['err']:.1e}) | "
    tms.sort()
    line += f"hipBLASLt med {flops / tms[len(tms) // 2] / 1e9:7.0f} TF"
    print(line, flush=True)
