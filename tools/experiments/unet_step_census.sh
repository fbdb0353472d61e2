#!/bin/bash
# kernels of ONE steady-state VDM-UNet train step (between two optimizer launches): counts and time per kernel name
export TMPDIR=/tmp
O=gpurun_out/unet_census; rm -rf $O
WHICH=unet_train timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O -- python3 tools/secondary_bench.py > gpurun_out/unet_census.log 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/unet_census/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
opt = [i for i, r in enumerate(rows) if "clip_adamw_ema" in r["Kernel_Name"]]
print("optimizer launches:", len(opt))
a, b = opt[-2], opt[-1]
step = rows[a + 1:b + 1]
t0, t1 = int(step[0]["Start_Timestamp"]), int(step[-1]["End_Timestamp"])
acc = collections.defaultdict(lambda: [0, 0])
busy = 0
for r in step:
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    nm = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:70]
    acc[nm][0] += 1; acc[nm][1] += d; busy += d
print(f"step wall {1e-6 * (t1 - t0):.2f} ms, kernels {len(step)}, busy {1e-6 * busy:.2f} ms, idle {1e-6 * (t1 - t0 - busy):.2f} ms")
for nm, (n, d) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:30]:
    print(f"{nm:70s} {n:5d} x {1e-3 * d / n:8.1f} us = {1e-6 * d:7.2f} ms")
# gaps: the largest idle intervals
gaps = sorted(((int(step[i + 1]["Start_Timestamp"]) - int(step[i]["End_Timestamp"]), step[i]["Kernel_Name"][:50], step[i + 1]["Kernel_Name"][:50]) for i in range(len(step) - 1)), reverse=True)[:8]
for g_, x, y in gaps: print(f"gap {1e-3 * g_:8.1f} us  after {x}  before {y}")
PY
rm -rf $O
