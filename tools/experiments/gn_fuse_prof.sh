#!/bin/bash
# per-kernel times of the UNet sampling loop with / without the fused GroupNorm statistics
export TMPDIR=/tmp
export K=8
O=gpurun_out
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/gnf_fused -- python3 tools/unet_bench.py > $O/gnf_fused.log 2>&1
export BSI_UNET_NO_GN_FUSE=1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/gnf_unfused -- python3 tools/unet_bench.py > $O/gnf_unfused.log 2>&1
for d in gnf_fused gnf_unfused; do echo "== $d"; python tools/kstats.py $(find $O/$d -name "*kernel_stats.csv" | head -1) | head -8; done
