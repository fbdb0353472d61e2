#!/bin/bash
# How many __amd_rocclr_copyBuffer (hipMemcpyAsync device-to-device) launches does ONE DPTrainer step issue?  Two profiles that differ
# in the number of steps only.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for S in 1 9; do
  B=64 STEPS=$S timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/cc$S -- python3 tools/train_profile.py > /dev/null 2>&1
  python3 - <<PY
import csv, glob
f = glob.glob("gpurun_out/cc$S/*/*_kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if "copyBuffer" in r["Name"] or "FillFunctor" in r["Name"] or "fillBuffer" in r["Name"]:
        print("steps $S:", r["Name"][:60], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), "us")
PY
  rm -rf gpurun_out/cc$S
done
