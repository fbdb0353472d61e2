#!/bin/bash
# timing ablations of the single-sweep attention backward: libraries that differ only in attention_bwd_x.o (-DATTN_ABL=bits), built HERE (CPU box)
# with `bash tools/experiments/attn_bwd_ablate.sh build`, timed on the GPU box with `... run`
cd "$(dirname "$0")/../.."
VARIANTS="${VARIANTS:-0 1 2 4 8 16 32 64 3 7 23 31}"
if [ "$1" = build ]; then
  mkdir -p tools/experiments/build/abl
  OTHERS=$(ls bsi_amd/lib/*.o | grep -v attention_bwd_x.o)
  for v in $VARIANTS; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -DATTN_ABL=$v -c bsi_amd/csrc/attention_bwd_x.hip -o tools/experiments/build/abl/x_$v.o &&
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OTHERS tools/experiments/build/abl/x_$v.o -o tools/experiments/build/abl/libbsi_hip_$v.so && rm tools/experiments/build/abl/x_$v.o &
    while [ $(jobs -r | wc -l) -ge 4 ]; do sleep 1; done
  done; wait; ls tools/experiments/build/abl
else
  mkdir -p gpurun_out/r4x
  for rep in 1 2; do
    for v in $VARIANTS; do
      TAG="abl=$v" BSI_HIP_LIB=$PWD/tools/experiments/build/abl/libbsi_hip_$v.so python tools/experiments/attn_bwd_time.py 2>&1 | grep median
    done
    BSI_ATTN_BWD_TWO_PASS=1 TAG="two-pass" python tools/experiments/attn_bwd_time.py 2>&1 | grep median
  done | tee gpurun_out/r4x/ablate.txt
fi
