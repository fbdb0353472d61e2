#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_ops.py -q -m gpu -x -k "attention or colsum" 2>&1 | tail -2
timeout 1800 python -m pytest tests/test_hip_dit.py tests/test_hip_dp_one_gpu.py tests/test_hip_fullsize_properties.py -q -m gpu -x 2>&1 | grep -E "passed|failed|rror" | tail -3
for i in 1 2 3; do STEPS=6 python tools/train_profile.py 2>&1 | tail -1; BSI_TRAIN_FUSED_BIAS=1 STEPS=6 python tools/train_profile.py 2>&1 | tail -1 | sed 's/^/all riders: /'; done
TAG="single sweep" python tools/experiments/attn_bwd_time.py | grep median
