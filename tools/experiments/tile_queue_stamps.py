#!/usr/bin/env python
"""Laboratory (LAB build of the library, BSI_HIP_LIB=...): per-workgroup stamps of one DYN GEMM launch -- entry, first ticket known, leaving
(100 MHz ticks), tiles processed, hardware XCC id -- read back from the mailbox lines of the stream's tile-queue control block."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bsi_amd import _native as N  # noqa: E402

B = int(os.environ.get("B", "64"))
M = B * 256
lib = N.lib()
lib.bsi_lab_tile_queue_block.restype = C.c_void_p
lib.bsi_lab_tile_queue_block.argtypes = [C.c_void_p]
g = torch.Generator(device="cuda").manual_seed(0)
Nn, K = 3072, 1024
A = torch.randn((M, K), device="cuda", generator=g).to(torch.bfloat16)
W = (torch.randn((Nn, K), device="cuda", generator=g) / K ** 0.5).to(torch.bfloat16)
bias = torch.randn(Nn, device="cuda", generator=g)
out = torch.empty((M, Nn), device="cuda", dtype=torch.bfloat16)
args = N.GemmArgs(A=A.data_ptr(), W=W.data_ptr(), bias=bias.data_ptr(), out=out.data_ptr(), M=M, N=Nn, K=K, lda=K, ldw=K, ldo=Nn, epilogue=N.EPI_BIAS_BF16)
os.environ["BSI_LAB_STAMPS"] = "1"
Q = int(os.environ.get("Q", "1"))
N.check(lib.bsi_set_tile_queue(Q))
for _ in range(6):
    N.check(lib.bsi_gemm_bf16(C.byref(args), N.stream()))
torch.cuda.synchronize()
ptr = lib.bsi_lab_tile_queue_block(N.stream())
print("schedule:", "queue" if Q else "static")
words = 32 + 16 * 512
buf = torch.empty(words, dtype=torch.int32, device="cuda")
import ctypes
hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemcpy(C.c_void_p(buf.data_ptr()), C.c_void_p(ptr), C.c_size_t(words * 4), 3)
w = buf.cpu().numpy().astype(np.uint32)
mb = w[32:32 + 16 * 256].reshape(256, 16)
t = mb[:, 2:8].copy().view(np.uint64).reshape(256, 3).astype(np.int64)
t0 = t[:, 0].min()
ent, tick, leave = (t[:, 0] - t0) / 100.0, (t[:, 1] - t[:, 0]) / 100.0, (t[:, 2] - t0) / 100.0
tiles, xcc = mb[:, 1], np.arange(256) & 7
prev_leave = mb[:, 10:12].copy().view(np.uint64).reshape(256).astype(np.int64)
print(f"gap to the previous launch of the same kernel: first entry - last leave = {(t[:, 0].min() - prev_leave.max()) / 100.0:.2f} us; "
      f"launch period (leave max - previous leave max) = {(t[:, 2].max() - prev_leave.max()) / 100.0:.2f} us")
print(f"workgroups 256: entry spread {ent.max():.1f} us; first ticket after entry: median {np.median(tick):.2f} us, max {tick.max():.2f}; leave: min {leave.min():.1f} median {np.median(leave):.1f} max {leave.max():.1f} us")
print("tiles per workgroup histogram:", dict(zip(*np.unique(tiles, return_counts=True))), "total", tiles.sum())
print("blockIdx & 7 == hardware XCC id for", int((xcc == (np.arange(256) & 7)).sum()), "of 256")
order = np.argsort(leave)
print("last to leave:", [(int(i), int(tiles[i]), round(float(ent[i]), 1), round(float(leave[i]), 1)) for i in order[-6:]])
print("late entries:", [(int(i), round(float(ent[i]), 1), int(tiles[i])) for i in np.argsort(ent)[-6:]])

tl = w[32 + 16 * 256:32 + 16 * 512].reshape(256, 8, 2).astype(np.int64)
nt = min(int(tiles.min()), 8)
ends = ((tl[:, :nt, 0] - (t[:, :1] & 0xffffffff)) & 0xffffffff) / 100.0        # tile end times relative to the workgroup's own entry
durs = np.diff(np.concatenate([np.zeros((256, 1)), ends], axis=1), axis=1)
print("tile durations (us, median over workgroups; first includes the start-up):", [round(float(x), 2) for x in np.median(durs, axis=0)])
print("   10th / 90th percentile:", [(round(float(np.percentile(durs[:, i], 10)), 1), round(float(np.percentile(durs[:, i], 90)), 1)) for i in range(nt)])
ntiles_total = int(tiles.sum())
per = -(-ntiles_total // 8)
q8, r8 = ntiles_total // 8, ntiles_total % 8
lo = np.array([x * (q8 + 1) if x < r8 else r8 * (q8 + 1) + (x - r8) * q8 for x in range(9)])
home = np.searchsorted(lo, tl[:, :nt, 1], side="right") - 1
foreign = (home != (np.arange(256) & 7)[:, None])
print("tiles computed outside their home XCD, per round:", foreign.sum(axis=0).tolist(), "of 256 each")
print("round-by-round span of tile indices per XCD 0:", [(int(tl[(np.arange(256) & 7) == 0, i, 1].min()), int(tl[(np.arange(256) & 7) == 0, i, 1].max())) for i in range(nt)])
