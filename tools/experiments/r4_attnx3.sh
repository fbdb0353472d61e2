#!/bin/bash
O=gpurun_out/r4x; mkdir -p $O
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
for cfg in "128 16" "300 4"; do set -- $cfg
  echo "== B=$1 H=$2"; B=$1 H=$2 LOCATE=1 SKIP_TORCH=1 timeout 600 python tools/experiments/attn_dropout_check.py 2>&1 | grep "bad pairs\|dqkv"
done
timeout 600 python -m pytest tests/test_hip_ops.py -q -m gpu -x -k "attention" 2>&1 | tail -1
for rep in 1 2 3; do
  TAG=x python tools/experiments/attn_bwd_time.py | grep median
  TAG=x-nodrop DROP=0 python tools/experiments/attn_bwd_time.py | grep median
  BSI_ATTN_BWD_TWO_PASS=1 TAG=two python tools/experiments/attn_bwd_time.py | grep median
  BSI_ATTN_BWD_TWO_PASS=1 TAG=two-nodrop DROP=0 python tools/experiments/attn_bwd_time.py | grep median
done
