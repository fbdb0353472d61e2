import os, sys, torch, ctypes as C
sys.path.insert(0, os.getcwd())
from bsi_amd import _native as N
lib = N.lib()
B, T, H, dh = 128, 256, 16, 64
d = H * dh
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
print(os.environ.get("BSI_ATTN_KC64", "kc256"), f"{ms*1e3:.1f} us  {4*B*H*T*T*dh/ms/1e9:.0f} TF  checksum {float(out.float().abs().mean()):.6f}")
