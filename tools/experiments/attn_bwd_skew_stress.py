"""Stress of the attention backward's two schedules (half a trip apart / lock step): R rounds of fresh random operands at B images x 16
heads with dropout words, every round both schedules, bit comparison of dQ / dK / dV.  A hand-over race would show as a differing round."""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from bsi_amd import _native as N
lib = N.lib()
R = int(os.environ.get("R", "40"))
T, H, dh = 256, 16, 64
d = H * dh
bad = 0
for r in range(R):
    B = (512, 70, 33, 257)[r % 4]
    p = (0.1, 0.0)[(r // 4) % 2]
    g = torch.Generator(device="cuda").manual_seed(1000 + r)
    qkv = (torch.randn((B, T, 3 * d), device="cuda", generator=g) * 1.2).to(torch.bfloat16)
    dout = torch.randn((B, T, d), device="cuda", generator=g).to(torch.bfloat16)
    out = torch.empty((B, T, d), device="cuda", dtype=torch.bfloat16)
    lse = torch.empty((B, H, T), device="cuda")
    mw = torch.zeros(B * H * 8192, dtype=torch.uint8, device="cuda") if p else None
    N.check(lib.bsi_attention_fwd_dropout(N.ptr(qkv), 3 * d, B, T, H, dh, N.ptr(out), d, N.ptr(lse), p, 11 + r, 2, N.ptr(mw) if p else None, N.stream()))
    res = []
    for on in (1, 0):
        lib.bsi_set_attention_bwd_skew(on)
        dqkv = torch.full((B, T, 3 * d), float("nan"), dtype=torch.bfloat16, device="cuda")
        N.check(lib.bsi_attention_bwd_dropout(N.ptr(qkv), 3 * d, N.ptr(out), N.ptr(dout), d, N.ptr(lse), B, T, H, dh, N.ptr(dqkv), 3 * d, p, 11 + r, 2,
                                              N.ptr(mw) if p else None, N.stream()))
        res.append(dqkv)
    torch.cuda.synchronize()
    same = torch.equal(res[0].view(torch.int16), res[1].view(torch.int16)) and bool(torch.isfinite(res[0].float()).all())
    bad += not same
    if not same:
        print(f"round {r}: B={B} p={p}: schedules differ in {int((res[0].view(torch.int16) != res[1].view(torch.int16)).sum())} values")
lib.bsi_set_attention_bwd_skew(1)
print(f"{R} rounds, {bad} differing")
sys.exit(1 if bad else 0)
