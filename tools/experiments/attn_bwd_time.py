"""Time the attention backward (DiT geometry, dropout words) through bsi_attention_bwd_dropout: B x H pairs, median of N launches."""
import ctypes as C, os, sys, statistics, torch
sys.path.insert(0, os.getcwd())
from bsi_amd import _native as N
lib = N.lib()
B, T, H, dh = int(os.environ.get("B", "512")), 256, int(os.environ.get("H", "16")), 64
d = H * dh
torch.manual_seed(0)
qkv = torch.randn((B, T, 3 * d), device="cuda").to(torch.bfloat16)
out = torch.randn((B, T, d), device="cuda").to(torch.bfloat16)
dout = torch.randn((B, T, d), device="cuda").to(torch.bfloat16)
lse = torch.randn((B, H, T), device="cuda") + 6.0
mw = torch.randint(0, 256, (B * H * 8192,), dtype=torch.uint8, device="cuda")
dqkv = torch.zeros((B, T, 3 * d), device="cuda", dtype=torch.bfloat16)
drop = int(os.environ.get("DROP", "1"))
def go():
    N.check(lib.bsi_attention_bwd_dropout(N.ptr(qkv), 3 * d, N.ptr(out), N.ptr(dout), d, N.ptr(lse), B, T, H, dh, N.ptr(dqkv), 3 * d,
                                          0.1 if drop else 0.0, 1234, 0, N.ptr(mw) if drop else None, N.stream()))
for _ in range(3): go()
torch.cuda.synchronize()
ts = []
for _ in range(int(os.environ.get("N", "12"))):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); go(); e1.record(); e1.synchronize()
    ts.append(1e3 * e0.elapsed_time(e1))
print(f"{os.environ.get('TAG', '')} B={B} H={H} drop={drop}: median {statistics.median(ts):.1f} us  min {min(ts):.1f}")
