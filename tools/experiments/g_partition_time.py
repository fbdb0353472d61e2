#!/usr/bin/env python
"""The engine's GEMMs and attention at a half batch (256 images) with a grid of 256 - h workgroups: on the unmasked stream (the
h CUs simply stay empty) against the CU-masked G stream of a pair (they are unreachable).  Nothing runs beside them."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bsi_amd import _native as N  # noqa: E402
from bsi_amd.models.dit import cu_pair_handle  # noqa: E402

dev = torch.device("cuda:0")
lib = N.lib()
B, T, d = int(os.environ.get("IMAGES", "256")), 256, 1024
M = B * T
g = torch.Generator(dev).manual_seed(0)
rb = lambda *s: (0.5 * torch.randn(s, device=dev, generator=g)).bfloat16()  # noqa: E731
a1, a4 = rb(M, d), rb(M, 4 * d)
w = {"qkv": rb(3 * d, d), "out": rb(d, d), "fc1": rb(4 * d, d), "fc2": rb(d, 4 * d)}
bias = torch.zeros(4 * d, device=dev)
o = torch.empty((M, 4 * d), dtype=torch.bfloat16, device=dev)
qkv = rb(M, 3 * d)


def gemm(name, s):
    K = 4 * d if name == "fc2" else d
    Nn = w[name].shape[0]
    ga = N.GemmArgs()
    ga.A, ga.W, ga.bias, ga.out = (a4 if name == "fc2" else a1).data_ptr(), w[name].data_ptr(), bias.data_ptr(), o.data_ptr()
    ga.M, ga.N, ga.K, ga.lda, ga.ldw, ga.ldo = M, Nn, K, K, K, Nn
    ga.epilogue = N.EPI_BIAS_GELU_BF16 if name == "fc1" else N.EPI_BIAS_BF16
    N.check(lib.bsi_gemm_bf16(C.byref(ga), s))


def attn(_, s):
    N.check(lib.bsi_attention_fwd(N.ptr(qkv), 3 * d, B, T, 16, 64, N.ptr(a1), d, s))


def timed(fn, name, s, n=int(os.environ.get("N", "300"))):
    ts = torch.cuda.ExternalStream(s.value, device=dev) if s.value else torch.cuda.current_stream()
    sp = s if s.value else N.stream()
    torch.cuda.synchronize()
    with torch.cuda.stream(ts):
        for _ in range(n // 3):  # let the clock settle under this load
            fn(name, sp)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n):
            fn(name, sp)
        b.record()
    b.synchronize()
    return a.elapsed_time(b) / n * 1e3


ops = [("qkv", gemm), ("out", gemm), ("fc1", gemm), ("fc2", gemm), ("attention", attn)]
print(f"# us per launch, {B} images; 'open' = unmasked stream, grid 256 - h; 'mask' = G stream of a pair with h CUs masked out")
print(f"{'h':>3s} " + " ".join(f"{n + ' open':>13s} {n + ' mask':>13s}" for n, _ in ops))
for h in [int(v) for v in os.environ.get('HS', '0,8,16,24,32,40,48,64,0').split(',')]:
    row = []
    N.check(lib.bsi_set_cu_reserve(h))
    for name, fn in ops:
        t_open = timed(fn, name, C.c_void_p(0))
        t_mask = float("nan")
        if h:
            sg, sh, hc = C.c_void_p(), C.c_void_p(), C.c_int()
            N.check(lib.bsi_cu_pair_streams(cu_pair_handle(dev, h), C.byref(sg), C.byref(sh), C.byref(hc)))
            t_mask = timed(fn, name, sg)
        row.append(f"{t_open:13.1f} {t_mask:13.1f}")
    N.check(lib.bsi_set_cu_reserve(0))
    print(f"{h:3d} " + " ".join(row))
