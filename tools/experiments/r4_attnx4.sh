#!/bin/bash
O=gpurun_out/r4x; mkdir -p $O
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_hip_ops.py -q -m gpu -x -k "attention" --durations=5 2>&1 | tail -12
B=64 H=16 LOCATE=1 SKIP_TORCH=1 timeout 300 python tools/experiments/attn_dropout_check.py 2>&1 | grep "bad pairs\|dqkv"
timeout 1500 python -m pytest tests/test_hip_dit.py tests/test_hip_dp_one_gpu.py -q -m gpu -x 2>&1 | grep -E "passed|failed|Error" | tail -3
for i in 1 2; do STEPS=5 python tools/train_profile.py 2>&1 | tail -1; BSI_ATTN_BWD_TWO_PASS=1 STEPS=5 python tools/train_profile.py 2>&1 | tail -1 | sed 's/^/two-pass: /'; done
