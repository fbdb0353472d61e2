"""Time bsi_ln_gate_bwd (LayerNorm-modulate backward + gated residual backward, 18 bytes per element) at the DiT-L training shape:
B images x 256 tokens x 1024 channels; median of N launches and the HBM rate of its algorithmic bytes."""
import os, sys, statistics, torch
sys.path.insert(0, os.getcwd())
from bsi_amd import _native as N
lib = N.lib()
B, T, d = int(os.environ.get("B", "512")), 256, 1024
M = B * T
g = torch.Generator(device="cuda").manual_seed(0)
dxn = torch.randn((M, d), device="cuda", generator=g).to(torch.bfloat16)
x = torch.randn((M, d), device="cuda", generator=g)
stats = torch.stack([x.mean(1), 1.0 / x.std(1)], 1).contiguous()
mod = torch.randn((B, 6 * d), device="cuda", generator=g) * 0.1
dmod = torch.zeros_like(mod)
dX = torch.randn((M, d), device="cuda", generator=g)
delta = torch.randn((M, d), device="cuda", generator=g).to(torch.bfloat16)
dd = torch.empty((M, d), device="cuda", dtype=torch.bfloat16)
def go():
    N.check(lib.bsi_ln_gate_bwd(N.ptr(dxn), N.ptr(x), N.ptr(stats), mod.data_ptr() + 4 * d, 6 * d, dmod.data_ptr(), dmod.data_ptr() + 4 * d, 6 * d,
                                N.ptr(dX), N.ptr(delta), mod.data_ptr() + 8 * d, 6 * d, dmod.data_ptr() + 8 * d, 6 * d, N.ptr(dd), M, d, T, N.stream()))
for _ in range(3): go()
torch.cuda.synchronize()
ts = []
for _ in range(int(os.environ.get("N", "20"))):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); go(); e1.record(); e1.synchronize()
    ts.append(1e3 * e0.elapsed_time(e1))
med = statistics.median(ts)
print(f"{os.environ.get('TAG', '')} B={B}: median {med:.1f} us  min {min(ts):.1f}  {18.0 * M * d / med / 1e6:.2f} TB/s")
