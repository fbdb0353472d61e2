#!/bin/bash
# Bias gradients relocated out of the weight-gradient GEMMs: parity tests, then the train step A/B (BSI_TRAIN_FUSED_BIAS=1 = before)
mkdir -p gpurun_out/r4bias
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_ops.py -q -m gpu -x -k "column_sums or colsum or ln_gate or training_epilogues" > gpurun_out/r4bias/ops.txt 2>&1
tail -3 gpurun_out/r4bias/ops.txt
timeout 1500 python -m pytest tests/test_hip_dit.py tests/test_hip_dp_one_gpu.py -q -m gpu -x > gpurun_out/r4bias/dit.txt 2>&1
tail -5 gpurun_out/r4bias/dit.txt
for i in 1 2; do
  BSI_TRAIN_FUSED_BIAS=1 STEPS=5 python tools/train_profile.py 2>&1 | tail -1 | sed 's/^/fused: /'
  STEPS=5 python tools/train_profile.py 2>&1 | tail -1 | sed 's/^/relocated: /'
done | tee gpurun_out/r4bias/ab.txt
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4bias/prof -- python3 tools/train_profile.py > gpurun_out/r4bias/prof.log 2>&1
python tools/kstats.py $(find gpurun_out/r4bias/prof -name "*kernel_stats.csv" | head -1) 40 > gpurun_out/r4bias/prof_summary.txt 2>&1 || true
