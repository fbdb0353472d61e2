#!/bin/bash
# single-sweep attention backward (attention_bwd_x.hip): parity, then timing against the two-pass kernel
O=gpurun_out/r4x; mkdir -p $O
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_hip_ops.py -q -m gpu -x -k "attention" > $O/ops.txt 2>&1; tail -3 $O/ops.txt
B=4 timeout 300 python tools/experiments/attn_dropout_check.py > $O/check_b4.txt 2>&1; tail -6 $O/check_b4.txt
B=200 timeout 300 python tools/experiments/attn_dropout_check.py > $O/check_b200.txt 2>&1; tail -6 $O/check_b200.txt
timeout 1200 python -m pytest tests/test_hip_dit.py -q -m gpu -x -k "dropout or train or grad" > $O/dit.txt 2>&1; grep -E "passed|failed" $O/dit.txt | tail -2
for arm in x two; do
  if [ $arm = two ]; then export BSI_ATTN_BWD_TWO_PASS=1; else unset BSI_ATTN_BWD_TWO_PASS; fi
  STEPS=3 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$arm -- python3 tools/train_profile.py > $O/prof_$arm.log 2>&1
  f=$(find $O/prof_$arm -name "*kernel_stats.csv" | head -1)
  echo "== $arm: $(grep 'ms/step' $O/prof_$arm.log)"; python tools/kstats.py $f 40 | grep -i "attention"
  cp $f $O/kernel_stats_$arm.csv; rm -rf $O/prof_$arm
done 2>&1 | tee $O/ab.txt
