"""Timing of the attention forward kernel at the DiT shapes (B x 16 heads x 256 tokens x 64): run twice, once with
BSI_ATTN_CHUNKED=1 (every 64-key chunk staged separately) for an A/B in two processes on the same box."""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from bsi_amd import _native as N
lib = N.lib()
T, H, dh = 256, 16, 64
d = H * dh
for B in (int(os.environ.get("B", "256")),):
    qkv = torch.randn((B, T, 3 * d), device="cuda").to(torch.bfloat16)
    out = torch.empty((B, T, d), device="cuda", dtype=torch.bfloat16)
    for _ in range(3):
        N.check(lib.bsi_attention_fwd(N.ptr(qkv), 3 * d, B, T, H, dh, N.ptr(out), d, N.stream()))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        N.check(lib.bsi_attention_fwd(N.ptr(qkv), 3 * d, B, T, H, dh, N.ptr(out), d, N.stream()))
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 50
    byts = B * T * 4 * d * 2
    print("chunked" if os.environ.get("BSI_ATTN_CHUNKED") else "whole K/V", f"B={B}: {ms*1e3:.1f} us  {4*B*H*T*T*dh/ms/1e9:.0f} TF  "
          f"{byts/ms/1e9:.2f} TB/s  checksum {float(out.float().abs().mean()):.6f}")
