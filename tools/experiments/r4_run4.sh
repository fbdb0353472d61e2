cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r4; mkdir -p $O
timeout 1200 python -m pytest tests/test_hip_dit.py tests/test_hip_ops.py tests/test_hip_parity_r2.py tests/test_hip_fullsize_properties.py tests/test_hip_dp_one_gpu.py -m gpu -x -q 2>&1 | grep -v "^PARITY\|^BOUND\|RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -4
for B in 64 512; do B=$B STEPS=5 timeout 600 python tools/train_profile.py 2>&1 | tail -1; done
B=64 STEPS=4 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tp64 -- python3 tools/train_profile.py > /dev/null 2>&1
cp $O/tp64/*/*_kernel_stats.csv $O/train_step_kernel_stats_b64_splitk.csv; rm -rf $O/tp64
python3 - <<PY
import csv
rows=list(csv.DictReader(open('gpurun_out/r4/train_step_kernel_stats_b64_splitk.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print("B=64 total kernel ms per step", tot/1e6/5)
for r in rows[:14]:
    print(r['Name'][:64].replace('(anonymous namespace)::',''), r['Calls'], round(float(r['AverageNs'])/1e3,1), r['Percentage'])
PY
