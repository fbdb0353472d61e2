cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r4; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_dit.py tests/test_hip_ops.py tests/test_hip_parity_r2.py tests/test_hip_fullsize_properties.py -m gpu -x -q 2>&1 | tail -12 > $O/suite_attn_tail.txt; tail -4 $O/suite_attn_tail.txt
B=512 STEPS=3 timeout 600 python tools/train_profile.py 2>&1 | tail -2
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/train_prof -- python3 tools/train_profile.py > /dev/null 2>&1
cp $O/train_prof/*/*_kernel_stats.csv $O/train_step_kernel_stats_a.csv; rm -rf $O/train_prof
head -14 $O/train_step_kernel_stats_a.csv | cut -c1-150
