#!/bin/bash
# GroupNorm backward: resident kernel (default) against the streaming one (BSI_GN_BWD_STREAM=1) inside the UNet train step, per launch.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for m in 1 0; do
  if [ $m = 1 ]; then export BSI_GN_BWD_STREAM=1; else unset BSI_GN_BWD_STREAM; fi
  WHICH=unet_train timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/gn$m -- python3 tools/secondary_bench.py > /dev/null 2>&1
  python3 - <<PY
import csv, glob
f = glob.glob("gpurun_out/gn$m/*/*_kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if "groupnorm_bwd" in r["Name"]:
        print("streaming" if $m else "resident ", r["Name"][:70], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), "us")
PY
  rm -rf gpurun_out/gn$m
done
