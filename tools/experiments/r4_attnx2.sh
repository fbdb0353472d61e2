#!/bin/bash
O=gpurun_out/r4x; mkdir -p $O
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
for cfg in "64 16" "128 16" "300 4" "512 16"; do set -- $cfg
  echo "== B=$1 H=$2"; B=$1 H=$2 LOCATE=1 SKIP_TORCH=1 timeout 600 python tools/experiments/attn_dropout_check.py 2>&1 | grep -v amdgpu.ids
done | tee $O/check_many.txt
timeout 600 python -m pytest tests/test_hip_ops.py -q -m gpu -x -k "attention" > $O/ops.txt 2>&1; tail -1 $O/ops.txt
for arm in x two; do
  if [ $arm = two ]; then export BSI_ATTN_BWD_TWO_PASS=1; else unset BSI_ATTN_BWD_TWO_PASS; fi
  STEPS=3 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$arm -- python3 tools/train_profile.py > $O/prof_$arm.log 2>&1
  f=$(find $O/prof_$arm -name "*kernel_stats.csv" | head -1)
  echo "== $arm: $(grep 'ms/step' $O/prof_$arm.log)"; python tools/kstats.py $f 40 | grep -i "attention"
  cp $f $O/kernel_stats_$arm.csv; rm -rf $O/prof_$arm
done 2>&1 | tee $O/ab.txt
