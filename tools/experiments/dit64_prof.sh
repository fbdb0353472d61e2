#!/bin/bash
# per-kernel times of DiT-L/4 (64x64, patch 4) sampling
export TMPDIR=/tmp
O=gpurun_out/dit64
rm -rf $O
WHICH=dit64 DIT64_K=8 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 tools/secondary_bench.py > gpurun_out/dit64.log 2>&1
python tools/kstats.py $(find $O -name "*kernel_stats.csv" | head -1) | head -14
