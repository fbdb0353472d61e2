#!/usr/bin/env python
"""A/B of the CU-partitioned stream pair (bsi_dit_forward_pair) against the one-stream engine on the headline workload
(DiT-L/2, 3x32x32, B images per call, k steps): interleaved arms, same generator state for every arm, the samples compared BIT FOR
BIT with the one-stream arm, and the shader clock seen by a one-wave probe kernel spinning beside each arm.

    python tools/experiments/cu_pair_ab.py [--batch 512] [--k 16] [--reps 3] [--h 16,24,32,40,48] [--queue 0,1]
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from bsi_amd import BSI, Discretization  # noqa: E402
from bsi_amd import _native as N  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=512)
    ap.add_argument("--k", type=int, default=16)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--h", default="16,24,32,40,48")
    ap.add_argument("--queue", default="0,1")
    ap.add_argument("--attn", default="g,h")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    model, shape = bench.build_model(dev)
    bsi = BSI(model, data_shape=shape, lambda_0=1e-2, alpha_M=1e6, alpha_R=2e6, k=a.k, preconditioning="edm",
              discretization=Discretization.image_8bit()).to(dev)
    lib = N.lib()
    arms = [("one_stream", None, 0)]
    for q in [int(v) for v in a.queue.split(",")]:
        if q:
            arms.append(("one_stream_queue", None, 1))
        for h in [int(v) for v in a.h.split(",")]:
            for at in a.attn.split(","):
                arms.append((f"pair_h{h}_attn_{at}_queue{q}", (h, 1 if at == "h" else 0), q))
    probe_stream = torch.cuda.Stream(device=dev)
    nprobe = 8
    clk = torch.zeros((nprobe, 2), dtype=torch.int64, device=dev)

    def run(pair, queue):
        model.cu_pair = pair
        N.check(lib.bsi_set_tile_queue(queue))
        g = torch.Generator(dev).manual_seed(1234)
        torch.cuda.synchronize()
        clk.zero_()
        t0 = time.perf_counter()
        with torch.cuda.stream(probe_stream):  # a one-wave kernel per 100 ms beside the chain: shader cycles per 100 MHz tick
            for i in range(nprobe):
                N.check(lib.bsi_clock_probe(N.ptr(clk[i]), 100000, C.c_void_p(probe_stream.cuda_stream)))
        with torch.no_grad():
            out = bsi.sample(a.batch, g)
        ev = torch.cuda.Event()
        ev.record()
        ev.synchronize()
        dt = time.perf_counter() - t0
        torch.cuda.synchronize()
        c = clk.cpu().double()
        n_in = max(1, min(nprobe, int(dt / 0.1)))  # probes that ran wholly beside the chain
        mhz = [float(c[i, 0] / c[i, 1] * 100.0) for i in range(nprobe)]
        return out, dt, mhz[:n_in]

    ref, _, _ = run(None, 0)  # warm-up + the reference bits
    for name, pair, q in arms[1:]:  # warm-up of every arm (stream creation, queue pool)
        run(pair, q)
    res = {n: {"s": [], "mhz": [], "identical": True} for n, _, _ in arms}
    for r in range(a.reps):
        for name, pair, q in arms:
            out, dt, mhz = run(pair, q)
            res[name]["s"].append(dt)
            res[name]["mhz"].append(sum(mhz) / len(mhz))
            res[name]["identical"] &= bool(torch.equal(out, ref))
    base = sorted(res["one_stream"]["s"])[len(res["one_stream"]["s"]) // 2]
    print(f"# DiT-L/2 BSI.sample, {a.batch} images, k = {a.k} ({a.k + 1} evaluations), {a.reps} interleaved repetitions; median seconds per "
          "call, images/s scaled to k = 128, shader MHz = one-wave probe beside the chain (mean of the 100 ms windows inside the call)")
    for name, _, _ in arms:
        s = sorted(res[name]["s"])
        med = s[len(s) // 2]
        mh = sum(res[name]["mhz"]) / len(res[name]["mhz"])
        print(f"{name:34s} {med:8.4f} s  ({min(s):.4f}..{max(s):.4f})  {base / med:6.3f}x  {a.batch / (med * 129 / (a.k + 1)):7.2f} img/s@k128  "
              f"{mh:7.1f} MHz  bit-identical={res[name]['identical']}")
    print(json.dumps({n: {"median_s": sorted(v["s"])[len(v["s"]) // 2], "mhz": sum(v["mhz"]) / len(v["mhz"]), "identical": v["identical"]}
                      for n, v in res.items()}))


if __name__ == "__main__":
    main()
