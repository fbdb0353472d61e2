"""Attention of the DiT geometry (512 images x 16 heads, 256 tokens, head dim 64, bf16) forward and backward: this library's kernels against
torch's F.scaled_dot_product_attention (the vendor flash-attention kernels on ROCm) -- a yardstick, same FLOPs and tensor sizes."""
import os, sys, statistics, torch
import torch.nn.functional as F
sys.path.insert(0, os.getcwd())
from bsi_amd import _native as N
lib = N.lib()
B, H, T, dh = int(os.environ.get("B", "512")), 16, 256, 64
d = H * dh
dev = torch.device("cuda")
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return statistics.median(ts)
torch.manual_seed(0)
qkv = torch.randn((B, T, 3 * d), device=dev).to(torch.bfloat16)
dout = torch.randn((B, T, d), device=dev).to(torch.bfloat16)
out = torch.empty((B, T, d), device=dev, dtype=torch.bfloat16)
lse = torch.empty((B, H, T), device=dev)
dqkv = torch.empty((B, T, 3 * d), device=dev, dtype=torch.bfloat16)
f_ours = timeit(lambda: N.check(lib.bsi_attention_fwd_lse(N.ptr(qkv), 3 * d, B, T, H, dh, N.ptr(out), d, N.ptr(lse), N.stream())))
b_ours = timeit(lambda: N.check(lib.bsi_attention_bwd(N.ptr(qkv), 3 * d, N.ptr(out), N.ptr(dout), d, N.ptr(lse), B, T, H, dh, N.ptr(dqkv), 3 * d, N.stream())))
q, k, v = (x.reshape(B, T, H, dh).transpose(1, 2).contiguous().requires_grad_(True) for x in qkv.split(d, dim=-1))
go = dout.reshape(B, T, H, dh).transpose(1, 2).contiguous()
f_vend = timeit(lambda: F.scaled_dot_product_attention(q, k, v))
o = F.scaled_dot_product_attention(q, k, v)
def bwd():
    torch.autograd.grad(o, (q, k, v), go, retain_graph=True)
b_vend = timeit(bwd)
fl_f, fl_b = 4.0 * B * H * T * T * dh, 10.0 * B * H * T * T * dh
ref = o.transpose(1, 2).reshape(B, T, d).float()
print(f"forward : ours {f_ours:7.1f} us = {fl_f / f_ours / 1e6:5.0f} TFLOP/s | torch SDPA {f_vend:7.1f} us = {fl_f / f_vend / 1e6:5.0f} TFLOP/s   (max diff {float((out.float() - ref).abs().max()):.1e})")
print(f"backward: ours {b_ours:7.1f} us = {fl_b / b_ours / 1e6:5.0f} TFLOP/s | torch SDPA {b_vend:7.1f} us = {fl_b / b_vend / 1e6:5.0f} TFLOP/s")
