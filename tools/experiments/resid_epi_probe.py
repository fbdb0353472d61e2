#!/usr/bin/env python
"""Would moving the gated residual update from ln_modulate into the out / fc2 GEMM epilogue pay?  Times, at 256 images:
 (a) GEMM + bias -> bf16 (variant 12) followed by resid_ln_modulate with the delta      (what the engine does)
 (b) GEMM with the GATE_RESID fp32 read-modify-write epilogue followed by ln_modulate without a delta"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bsi_amd import _native as N  # noqa: E402

dev = "cuda"
B = int(os.environ.get("B", "256"))
M, d = B * 256, 1024
lib = N.lib()
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn((M, d), device=dev, generator=g)
mod = torch.randn((B, 6 * d), device=dev, generator=g) * 0.1
h = torch.empty((M, d), device=dev, dtype=torch.bfloat16)


def timeit(fn, iters=10, rounds=5):
    fn(); fn()
    torch.cuda.synchronize()
    ms = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1) / iters * 1e3)
    ms.sort()
    return ms[len(ms) // 2]


for name, K in (("out", 1024), ("fc2", 4096)):
    A = torch.randn((M, K), device=dev, generator=g).to(torch.bfloat16)
    W = (torch.randn((d, K), device=dev, generator=g) / K ** 0.5).to(torch.bfloat16)
    bias = torch.randn(d, device=dev, generator=g)
    y = torch.empty((M, d), device=dev, dtype=torch.bfloat16)
    a1 = N.GemmArgs(A=A.data_ptr(), W=W.data_ptr(), bias=bias.data_ptr(), M=M, N=d, K=K, lda=K, ldw=K, ldo=d,
                    epilogue=N.EPI_BIAS_BF16, tokens=256)
    a1.out = y.data_ptr()
    a2 = N.GemmArgs(A=A.data_ptr(), W=W.data_ptr(), bias=bias.data_ptr(), M=M, N=d, K=K, lda=K, ldw=K, ldo=d,
                    epilogue=N.EPI_GATE_RESID, gate=mod.data_ptr(), gate_rows=B, gate_stride=6 * d, tokens=256)
    a2.out = x.data_ptr()
    fp = lambda t: C.c_void_p(t.data_ptr())
    off = lambda t, o: C.c_void_p(t.data_ptr() + 4 * o)

    def gemm_bf16():
        N.check(lib.bsi_gemm_bf16(C.byref(a1), N.stream()))

    def gemm_resid():
        N.check(lib.bsi_gemm_bf16(C.byref(a2), N.stream()))

    def ln_delta():
        N.check(lib.bsi_resid_ln_modulate(fp(x), M, d, 1e-5, N.ptr(y), fp(mod), off(mod, d), off(mod, 2 * d), B, 6 * d, 256,
                                          None, None, N.ptr(h), N.stream()))

    def ln_plain():
        N.check(lib.bsi_resid_ln_modulate(fp(x), M, d, 1e-5, None, None, off(mod, d), off(mod, 2 * d), B, 6 * d, 256,
                                          None, None, N.ptr(h), N.stream()))

    t = {k: timeit(f) for k, f in (("gemm_bf16", gemm_bf16), ("gemm_resid", gemm_resid), ("ln_delta", ln_delta), ("ln_plain", ln_plain))}
    ta = timeit(lambda: (gemm_bf16(), ln_delta()))
    tb = timeit(lambda: (gemm_resid(), ln_plain()))
    print(f"{name}: " + "  ".join(f"{k} {v:7.1f} us" for k, v in t.items()) + f"  | pair (a) {ta:7.1f} us  pair (b) {tb:7.1f} us", flush=True)
    x.normal_()
