"""Print nothing; run under rocprofv3 --kernel-trace to see which hipBLASLt kernels torch.matmul picks for the DiT-L
shapes (the yardstick's tile configuration is encoded in the kernel name)."""
import torch
M = 32768
for n, k in [(3072, 1024), (1024, 1024), (4096, 1024), (1024, 4096)]:
    a = torch.randn(M, k, device="cuda", dtype=torch.bfloat16)
    w = torch.randn(n, k, device="cuda", dtype=torch.bfloat16)
    for _ in range(3):
        (a @ w.t())
torch.cuda.synchronize()
