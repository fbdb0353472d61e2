"""Weight-gradient GEMM dW = dY^T X (both operands with the token index slow) at the DiT-L/2 training shapes: this library's gemm_tn2 kernel
(bsi_gemm_tn_bf16, slabs + reduction included) against the vendor library through torch.mm on the transposed view (bf16 output: a yardstick
for the kernel rate, not a drop-in -- the engine needs fp32 sums)."""
import os, sys, statistics, torch
sys.path.insert(0, os.getcwd())
from bsi_amd import _native as N
lib = N.lib()
M = int(os.environ.get("M", str(512 * 256)))
dev = torch.device("cuda")
def timeit(fn, n=8):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return statistics.median(ts)
for name, Nn, K in (("qkv", 3072, 1024), ("out", 1024, 1024), ("fc1", 4096, 1024), ("fc2", 1024, 4096)):
    dY = torch.randn((M, Nn), device=dev).to(torch.bfloat16)
    X = torch.randn((M, K), device=dev).to(torch.bfloat16)
    out = torch.empty((Nn, K), device=dev)
    ws = torch.empty(lib.bsi_gemm_tn_workspace_bytes(M, Nn, K), dtype=torch.uint8, device=dev)
    ours = timeit(lambda: N.check(lib.bsi_gemm_tn_bf16(N.ptr(dY), Nn, N.ptr(X), K, M, Nn, K, N.ptr(out), K, 0, N.ptr(ws), N.stream())))
    dYt = dY.t()
    vend = timeit(lambda: torch.mm(dYt, X))
    fl = 2.0 * M * Nn * K
    ref = torch.mm(dYt, X).float()
    err = float((out - ref).abs().max() / ref.abs().max())
    print(f"{name:4s} N={Nn} K={K} M={M}: gemm_tn2 {ours:8.1f} us = {fl / ours / 1e6:6.0f} TFLOP/s | torch.mm(dY^T, X) {vend:8.1f} us = {fl / vend / 1e6:6.0f} TFLOP/s   (max diff {err:.1e})")
