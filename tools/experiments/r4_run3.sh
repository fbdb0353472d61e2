cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r4; mkdir -p $O
python tools/experiments/dbg_attn_drop.py 2>&1 | tail -4
timeout 900 python -m pytest tests/test_hip_dit.py tests/test_hip_ops.py tests/test_hip_parity_r2.py -m gpu -x -q 2>&1 | grep -v "^PARITY\|^BOUND\|RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -4
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/train_prof -- python3 tools/train_profile.py > /dev/null 2>&1
cp $O/train_prof/*/*_kernel_stats.csv $O/train_step_kernel_stats_b.csv; rm -rf $O/train_prof
python3 - <<'PY'
import csv
tot=0
rows=list(csv.DictReader(open('gpurun_out/r4/train_step_kernel_stats_b.csv')))
for r in rows:
    n=r['Name']; tot+=float(r['TotalDurationNs'])
for r in rows[:16]:
    print(r['Name'][:70].replace('(anonymous namespace)::',''), r['Calls'], round(float(r['AverageNs'])/1e3,1), r['Percentage'])
print("total ms per step", tot/1e6/4)
PY
B=512 STEPS=5 timeout 600 python tools/train_profile.py 2>&1 | tail -1
