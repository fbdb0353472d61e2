"""Shader clock right after a burst of GEMMs (ours vs the hipBLASLt yardstick) on N(0,1) x U(+-1/sqrt(K)) operands:
shows which DVFS state each kernel drives the chip into (power-limited parts trade utilisation for clock)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bsi_amd import _native as N  # noqa: E402

lib = N.lib()
M = 32768
clk = torch.zeros(2, dtype=torch.int64, device="cuda")


def probe():
    N.check(lib.bsi_clock_probe(N.ptr(clk), 50, N.stream()))
    torch.cuda.synchronize()
    c = clk.cpu()
    return 100.0 * float(c[0]) / float(c[1])


print(f"idle clock {probe():.0f} MHz")
for name, Nn, K in [("qkv", 3072, 1024), ("fc1", 4096, 1024), ("fc2", 1024, 4096)]:
    A = torch.randn((M, K), device="cuda").to(torch.bfloat16)
    W = ((torch.rand((Nn, K), device="cuda") * 2 - 1) / K ** 0.5).to(torch.bfloat16)
    bias = torch.zeros(Nn, device="cuda")
    out = torch.empty((M, Nn), device="cuda", dtype=torch.bfloat16)
    args = N.GemmArgs(A=A.data_ptr(), W=W.data_ptr(), bias=bias.data_ptr(), M=M, N=Nn, K=K, lda=K, ldw=K, ldo=Nn,
                      epilogue=N.EPI_BIAS_BF16, out=out.data_ptr())
    fl = 2.0 * M * Nn * K
    for who in ("ours", "hipBLASLt", "ours", "hipBLASLt"):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        iters = 200
        e0.record()
        for _ in range(iters):
            if who == "ours":
                N.check(lib.bsi_gemm_bf16(C.byref(args), N.stream()))
            else:
                torch.matmul(A, W.t(), out=out)
        e1.record()
        mhz = probe()
        ms = e0.elapsed_time(e1) / iters
        print(f"{name} {who:10s} {fl / ms / 1e9:7.0f} TF   clock right after: {mhz:.0f} MHz")
