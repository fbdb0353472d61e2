"""Times the two attention-backward kernels on the DiT-L training shape (B x 16 heads, 256 tokens, dh 64, no dropout)."""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from bsi_amd import _native as N
lib = N.lib()
B, T, H, dh = int(os.environ.get("B", "512")), 256, 16, 64
d = H * dh
qkv = torch.randn((B, T, 3 * d), device="cuda").to(torch.bfloat16)
dout = torch.randn((B, T, d), device="cuda").to(torch.bfloat16)
out = torch.empty((B, T, d), device="cuda", dtype=torch.bfloat16)
lse = torch.empty((B, H, T), device="cuda")
dqkv = torch.empty((B, T, 3 * d), device="cuda", dtype=torch.bfloat16)
N.check(lib.bsi_attention_fwd_lse(N.ptr(qkv), 3 * d, B, T, H, dh, N.ptr(out), d, N.ptr(lse), N.stream()))
res = {}
for name, fn in (("resident", lib.bsi_attention_bwd), ("stream", lib.bsi_attention_bwd_long)):
    for _ in range(2):
        N.check(fn(N.ptr(qkv), 3 * d, N.ptr(out), N.ptr(dout), d, N.ptr(lse), B, T, H, dh, N.ptr(dqkv), 3 * d, N.stream()))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        N.check(fn(N.ptr(qkv), 3 * d, N.ptr(out), N.ptr(dout), d, N.ptr(lse), B, T, H, dh, N.ptr(dqkv), 3 * d, N.stream()))
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    res[name] = dqkv.float().clone()
    print(f"{name:9s} {ms*1e3:8.1f} us   {14*B*H*T*T*dh/ms/1e9:.0f} TF (7 products)")
print("max diff", float((res["resident"] - res["stream"]).abs().max()))
