#!/bin/bash
# per-kernel times of the VDM-UNet train step (batch 128)
export TMPDIR=/tmp
O=gpurun_out/unet_t
rm -rf $O
WHICH=unet_train timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 tools/secondary_bench.py > gpurun_out/unet_t.log 2>&1
python tools/kstats.py $(find $O -name "*kernel_stats.csv" | head -1) | head -24
