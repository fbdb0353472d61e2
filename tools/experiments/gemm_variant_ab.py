"""Variant 12 (every wave issues a quarter of every half-stage) against variant 13 (operand DMA split by wave group) of the bf16 GEMM:
bit-exact comparison of every epilogue at full, ragged and short shapes, and alternating timing at the DiT-L shapes (512 images)."""
import ctypes as C
import os, sys
import torch
sys.path.insert(0, os.getcwd())
from bsi_amd import _native as N
lib = N.lib()
g = torch.Generator(device="cuda").manual_seed(0)
EPIS = [("bias", N.EPI_BIAS_BF16), ("gelu", N.EPI_BIAS_GELU_BF16), ("silu", N.EPI_BIAS_SILU_BF16), ("dual", N.EPI_BIAS_GELU_DUAL),
        ("ggrad", N.EPI_MUL_GELUGRAD_BF16)]


def run(M, Nn, K, epi, variant, A, W, bias, aux, iters=0):
    out = torch.zeros((M, Nn), device="cuda", dtype=torch.bfloat16)
    out2 = torch.zeros((M, Nn), device="cuda", dtype=torch.bfloat16)
    args = N.GemmArgs(A=A.data_ptr(), W=W.data_ptr(), bias=bias.data_ptr(), M=M, N=Nn, K=K, lda=K, ldw=K, ldo=Nn, epilogue=epi, tokens=256,
                      out=out.data_ptr(), out2=out2.data_ptr(), aux=aux.data_ptr())
    N.check(lib.bsi_gemm_set_variant(variant))
    N.check(lib.bsi_gemm_bf16(C.byref(args), N.stream()))
    ms = None
    if iters:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            N.check(lib.bsi_gemm_bf16(C.byref(args), N.stream()))
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
    torch.cuda.synchronize()
    N.check(lib.bsi_gemm_set_variant(12))
    return out, out2, ms


bad = 0
for M in (4096, 1000, 300, 257, 131072 if os.environ.get("BIG") else 8192):
    for Nn, K in ((3072, 1024), (4096, 1024), (1024, 4096), (1024, 1024), (512, 128), (256, 64), (1152, 192)):
        A = torch.randn((M, K), device="cuda", generator=g).to(torch.bfloat16)
        W = (torch.randn((Nn, K), device="cuda", generator=g) / K ** 0.5).to(torch.bfloat16)
        bias = torch.randn(Nn, device="cuda", generator=g)
        aux = torch.randn((M, Nn), device="cuda", generator=g).to(torch.bfloat16)
        for name, epi in EPIS:
            a = run(M, Nn, K, epi, 12, A, W, bias, aux)
            b = run(M, Nn, K, epi, 13, A, W, bias, aux)
            ok = torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
            if not ok:
                bad += 1
                print(f"MISMATCH M={M} N={Nn} K={K} {name}: max diff {float((a[0].float() - b[0].float()).abs().max()):.3e}")
print("bit-exact comparison:", "ALL EQUAL" if bad == 0 else f"{bad} MISMATCHES")

M = 131072
for name, Nn, K, epi in [("qkv", 3072, 1024, N.EPI_BIAS_BF16), ("out", 1024, 1024, N.EPI_BIAS_BF16), ("fc1 gelu", 4096, 1024, N.EPI_BIAS_GELU_BF16),
                         ("fc2", 1024, 4096, N.EPI_BIAS_BF16), ("fc1 dual", 4096, 1024, N.EPI_BIAS_GELU_DUAL), ("ggrad", 1024, 4096, N.EPI_MUL_GELUGRAD_BF16)]:
    A = torch.randn((M, K), device="cuda", generator=g).to(torch.bfloat16)
    W = (torch.randn((Nn, K), device="cuda", generator=g) / K ** 0.5).to(torch.bfloat16)
    bias = torch.randn(Nn, device="cuda", generator=g)
    aux = torch.randn((M, Nn), device="cuda", generator=g).to(torch.bfloat16)
    for r in range(3):
        t12 = run(M, Nn, K, epi, 12, A, W, bias, aux, 10)[2]
        t13 = run(M, Nn, K, epi, 13, A, W, bias, aux, 10)[2]
        fl = 2.0 * M * Nn * K
        print(f"{name:9s} variant 12 {t12*1e3:7.1f} us {fl/t12/1e9:6.0f} TF | variant 13 {t13*1e3:7.1f} us {fl/t13/1e9:6.0f} TF | {100*(t12/t13-1):+.1f} %")
