#!/bin/bash
# A/B of the fused GroupNorm statistics (conv epilogue partials + streaming apply) against the reduce-then-normalise kernel:
# UNet sampling (k=32) and the UNet train step, interleaved, on one box.
mkdir -p gpurun_out
for r in 1 2; do
  K=32 python tools/unet_bench.py 2>&1 | tail -1 | sed 's/^/fused   /'
  BSI_UNET_NO_GN_FUSE=1 K=32 python tools/unet_bench.py 2>&1 | tail -1 | sed 's/^/unfused /'
done
WHICH=unet_train python tools/secondary_bench.py 2>&1 | tail -1 | sed 's/^/fused   /'
BSI_UNET_NO_GN_FUSE=1 WHICH=unet_train python tools/secondary_bench.py 2>&1 | tail -1 | sed 's/^/unfused /'
