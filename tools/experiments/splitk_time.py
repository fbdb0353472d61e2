"""Split-K latency path vs the ordinary kernel for the four DiT-L GEMM shapes at small M (us per launch)."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.getcwd())
from bsi_amd import _native as N
lib = N.lib()
dev = torch.device("cuda:0")
shapes = [("qkv", 3072, 1024, N.EPI_BIAS_BF16), ("out", 1024, 1024, N.EPI_BIAS_BF16), ("fc1", 4096, 1024, N.EPI_BIAS_GELU_BF16), ("fc2", 1024, 4096, N.EPI_BIAS_BF16)]
for M in (256, 512, 1024, 2048):
    line = f"M={M:5d}"
    for name, Nn, K, epi in shapes:
        A = torch.randn(M, K, device=dev).to(torch.bfloat16); W = torch.randn(Nn, K, device=dev).to(torch.bfloat16) / K ** 0.5
        b = torch.randn(Nn, device=dev); out = torch.empty(M, Nn, device=dev, dtype=torch.bfloat16)
        need = lib.bsi_gemm_splitk_workspace_bytes(M, Nn, K)
        ws = torch.empty(max(need, 16), dtype=torch.uint8, device=dev)
        a = N.GemmArgs(A=A.data_ptr(), W=W.data_ptr(), bias=b.data_ptr(), out=out.data_ptr(), M=M, N=Nn, K=K, lda=K, ldw=K, ldo=Nn, epilogue=epi)
        res = []
        for fn in (lambda: lib.bsi_gemm_bf16(C.byref(a), N.stream()), lambda: lib.bsi_gemm_bf16_ws(C.byref(a), N.ptr(ws), need, N.stream())):
            for _ in range(5): fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50): fn()
            e1.record(); torch.cuda.synchronize()
            res.append(e0.elapsed_time(e1) / 50 * 1e3)
        line += f"  {name}: {res[0]:5.1f} -> {res[1]:5.1f}{'' if need else ' (no split)'}"
    print(line, flush=True)
