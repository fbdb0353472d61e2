#!/bin/bash
# UNet sampling (k=32) with the slab / ring kernel choice varied (BSI_CONV_ABL: 1024 = ring for the fp32 epilogues, 2048 = ring for
# the FiLM epilogue, 256 = ring everywhere), then per-kernel times of the all-ring choice
for r in 1 2; do
  for abl in 0 1024 3072; do
    BSI_CONV_ABL=$abl K=32 python tools/unet_bench.py 2>&1 | tail -1 | sed "s/^/abl=$abl /"
  done
done
export TMPDIR=/tmp
for abl in 0 3072; do
BSI_CONV_ABL=$abl K=8 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/cc_$abl -- python3 tools/unet_bench.py > /dev/null 2>&1
echo "== abl $abl"; python tools/kstats.py $(find gpurun_out/cc_$abl -name "*kernel_stats.csv" | head -1) | head -7
done
