#!/usr/bin/env python
"""One BSI.sample call (k steps) of the headline model, for a rocprofv3 --kernel-trace timeline:
    BSI_CU_PAIR=32 rocprofv3 --kernel-trace --output-format csv -d out -- python3 tools/experiments/cu_pair_trace.py
    python tools/experiments/cu_pair_trace.py --analyze out        (per-kernel durations, per-queue busy time, overlap)"""
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0][:60]


def analyze(d):
    f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    rows = [r for r in rows if "Start_Timestamp" in r]
    for r in rows:
        r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    # the LAST sample call only: everything after the largest gap-free tail is fine -- take the last `tail` fraction
    rows.sort(key=lambda r: r["s"])
    t_end = rows[-1]["e"]
    win = float(os.environ.get("WIN_MS", "150")) * 1e6
    rows = [r for r in rows if r["s"] >= t_end - win]
    t0 = rows[0]["s"]
    span = (t_end - t0) / 1e6
    by, q = {}, {}
    for r in rows:
        k = short(r["Kernel_Name"])
        by.setdefault(k, []).append((r["e"] - r["s"]) / 1e3)
        q.setdefault(r["Queue_Id"], []).append(r)
    print(f"window {span:.1f} ms, {len(rows)} kernels")
    for k, v in sorted(by.items(), key=lambda kv: -sum(kv[1]))[:14]:
        print(f"  {k:60s} {len(v):5d} x {sum(v) / len(v):8.1f} us = {sum(v) / 1e3:8.2f} ms")
    for qid, rs in q.items():
        busy = sum(r["e"] - r["s"] for r in rs) / 1e6
        names = sorted({short(r["Kernel_Name"]) for r in rs})
        print(f"  queue {qid}: {len(rs)} kernels, busy {busy:.1f} ms = {100 * busy / span:.0f} % of the window; kernels: {', '.join(n[:28] for n in names[:6])}")
    if os.environ.get("DUMP"):
        for r in rows[: int(os.environ["DUMP"])]:
            print(f"    q{r['Queue_Id']} {(r['s'] - t0) / 1e3:10.1f} +{(r['e'] - r['s']) / 1e3:8.1f} us  grid {r.get('Grid_Size', '?'):>8s} {short(r['Kernel_Name'])}")


def main():
    import torch
    import bench
    from bsi_amd import BSI, Discretization
    dev = torch.device("cuda:0")
    k = int(os.environ.get("K", "2"))
    model, shape = bench.build_model(dev)
    bsi = BSI(model, data_shape=shape, lambda_0=1e-2, alpha_M=1e6, alpha_R=2e6, k=k, preconditioning="edm",
              discretization=Discretization.image_8bit()).to(dev)
    g = torch.Generator(dev).manual_seed(0)
    with torch.no_grad():
        for _ in range(2):
            bsi.sample(int(os.environ.get("B", "512")), g)
    torch.cuda.synchronize()


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--analyze":
        analyze(sys.argv[2])
    else:
        main()
