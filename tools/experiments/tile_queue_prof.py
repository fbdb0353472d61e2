#!/usr/bin/env python
"""Kernel-trace subject for the tile queue: the DiT-L/2 GEMM shapes at B images, static then queue, 20 launches each (run under
rocprofv3 --kernel-trace --stats: the two schedules are different template instances and show up as separate rows)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bsi_amd import _native as N  # noqa: E402

B = int(os.environ.get("B", "64"))
M = B * 256
lib = N.lib()
g = torch.Generator(device="cuda").manual_seed(0)
for name, Nn, K, epi in [("qkv", 3072, 1024, N.EPI_BIAS_BF16), ("fc1", 4096, 1024, N.EPI_BIAS_GELU_BF16)]:
    A = torch.randn((M, K), device="cuda", generator=g).to(torch.bfloat16)
    W = (torch.randn((Nn, K), device="cuda", generator=g) / K ** 0.5).to(torch.bfloat16)
    bias = torch.randn(Nn, device="cuda", generator=g)
    out = torch.empty((M, Nn), device="cuda", dtype=torch.bfloat16)
    args = N.GemmArgs(A=A.data_ptr(), W=W.data_ptr(), bias=bias.data_ptr(), out=out.data_ptr(), M=M, N=Nn, K=K, lda=K, ldw=K, ldo=Nn, epilogue=epi)
    for rnd in range(3):
        for q in (0, 1):
            N.check(lib.bsi_set_tile_queue(q))
            for _ in range(20):
                N.check(lib.bsi_gemm_bf16(C.byref(args), N.stream()))
            torch.cuda.synchronize()
N.check(lib.bsi_set_tile_queue(0))
