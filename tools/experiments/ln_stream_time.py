#!/usr/bin/env python
"""LayerNorm + modulate passes of the inference engine (256 images x 256 tokens x 1024) on the H partition of a CU pair and on the whole
chip: the one-row-per-wave kernel against the persistent prefetching one.  Both forms of the pass (first pass of a block: two pending
updates, row stored, 14 B per element; second pass: one update kept in registers, 8 B per element)."""
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
M, d = int(os.environ.get("IMAGES", "256")) * 256, 1024
g = torch.Generator(dev).manual_seed(0)
x0 = torch.randn((M, d), device=dev, generator=g)
da = torch.randn((M, d), device=dev, generator=g).bfloat16()
db = torch.randn((M, d), device=dev, generator=g).bfloat16()
mod = 0.1 * torch.randn((6, d), device=dev, generator=g)
out = torch.empty((M, d), dtype=torch.bfloat16, device=dev)


def launch(x, first, s):
    if first:
        N.check(lib.bsi_resid2_ln_modulate(N.ptr(x), M, d, 1e-5, N.ptr(da), N.ptr(mod[2]), N.ptr(db), N.ptr(mod[5]), 1, N.ptr(mod[0]),
                                           N.ptr(mod[1]), 1, 6 * d, 256, N.ptr(out), s))
    else:
        N.check(lib.bsi_resid2_ln_modulate(N.ptr(x), M, d, 1e-5, None, None, N.ptr(da), N.ptr(mod[2]), 0, N.ptr(mod[3]), N.ptr(mod[4]),
                                           1, 6 * d, 256, N.ptr(out), s))


def timed(first, stream_cus, s, n=20):
    N.check(lib.bsi_set_ln_stream_cus(stream_cus))
    ts = torch.cuda.ExternalStream(s.value, device=dev) if s.value else torch.cuda.current_stream()
    sp = s if s.value else N.stream()
    x = x0.clone()
    torch.cuda.synchronize()
    with torch.cuda.stream(ts):
        launch(x, first, sp)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n):
            launch(x, first, sp)
        b.record()
    b.synchronize()
    N.check(lib.bsi_set_ln_stream_cus(0))
    return a.elapsed_time(b) / n * 1e3, out.clone(), x


for first in (True, False):
    bytes_ = M * d * (14 if first else 8)
    print(f"# {'first pass (two updates, row stored, 14 B/el)' if first else 'second pass (one update, not stored, 8 B/el)'}: {bytes_ / 1e6:.0f} MB")
    t, ref, xr = timed(first, 0, C.c_void_p(0))
    print(f"  whole chip, one row per wave      {t:8.1f} us  {bytes_ / t / 1e6:6.2f} TB/s")
    t, o, xx = timed(first, 256, C.c_void_p(0))
    print(f"  whole chip, persistent            {t:8.1f} us  {bytes_ / t / 1e6:6.2f} TB/s  bit-identical={torch.equal(o, ref) and torch.equal(xx, xr)}")
    for h in (8, 16, 24, 32, 48):
        pair = cu_pair_handle(dev, h)
        sg, sh, hc = C.c_void_p(), C.c_void_p(), C.c_int()
        N.check(lib.bsi_cu_pair_streams(pair, C.byref(sg), C.byref(sh), C.byref(hc)))
        t0, o0, x0_ = timed(first, 0, sh)
        t1, o1, x1_ = timed(first, h, sh)
        print(f"  H = {h:2d} CUs: one row per wave {t0:8.1f} us {bytes_ / t0 / 1e3 / h:6.1f} GB/s/CU | persistent {t1:8.1f} us {bytes_ / t1 / 1e3 / h:6.1f} GB/s/CU "
              f"= {bytes_ / t1 / 1e6:5.2f} TB/s  bit-identical={torch.equal(o1, ref) and torch.equal(x1_, xr) and torch.equal(o0, ref)}")
