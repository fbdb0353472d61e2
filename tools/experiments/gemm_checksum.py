"""Bit-level fingerprint of the GEMM outputs at the DiT-L shapes (all epilogues the engines use, ragged M included): run under two
builds (BSI_HIP_LIB=...) to show that a source change left every output bit unchanged."""
import ctypes as C
import os, sys, hashlib
import torch
sys.path.insert(0, os.getcwd())
from bsi_amd import _native as N
lib = N.lib()
g = torch.Generator(device="cuda").manual_seed(0)
for M in (65536, 1000, 300):
    for name, Nn, K, epi in [("qkv", 3072, 1024, N.EPI_BIAS_BF16), ("fc1", 4096, 1024, N.EPI_BIAS_GELU_BF16), ("fc2", 1024, 4096, N.EPI_BIAS_BF16),
                             ("silu", 1024, 1024, N.EPI_BIAS_SILU_BF16), ("f32", 1024, 1024, N.EPI_BIAS_F32), ("dual", 4096, 1024, N.EPI_BIAS_GELU_DUAL),
                             ("ggrad", 1024, 4096, N.EPI_MUL_GELUGRAD_BF16)]:
        A = torch.randn((M, K), device="cuda", generator=g).to(torch.bfloat16)
        W = (torch.randn((Nn, K), device="cuda", generator=g) / K ** 0.5).to(torch.bfloat16)
        bias = torch.randn(Nn, device="cuda", generator=g)
        f32 = epi == N.EPI_BIAS_F32
        out = torch.zeros((M, Nn), device="cuda", dtype=torch.float32 if f32 else torch.bfloat16)
        out2 = torch.zeros((M, Nn), device="cuda", dtype=torch.bfloat16)
        aux = torch.randn((M, Nn), device="cuda", generator=g).to(torch.bfloat16)
        args = N.GemmArgs(A=A.data_ptr(), W=W.data_ptr(), bias=bias.data_ptr(), M=M, N=Nn, K=K, lda=K, ldw=K, ldo=Nn, epilogue=epi, tokens=256,
                          out=out.data_ptr(), out2=out2.data_ptr(), aux=aux.data_ptr())
        N.check(lib.bsi_gemm_bf16(C.byref(args), N.stream()))
        torch.cuda.synchronize()
        h = hashlib.sha256(out.cpu().view(torch.uint8).numpy().tobytes() + out2.cpu().view(torch.uint8).numpy().tobytes()).hexdigest()[:16]
        print(f"M={M:6d} {name:6s} {h}")
