cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r4; mkdir -p $O
for B in 64 128; do
B=$B STEPS=4 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tp$B -- python3 tools/train_profile.py > /dev/null 2>&1
cp $O/tp$B/*/*_kernel_stats.csv $O/train_step_kernel_stats_b$B.csv; rm -rf $O/tp$B
python3 - <<PY
import csv
rows=list(csv.DictReader(open('gpurun_out/r4/train_step_kernel_stats_b$B.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print("B=$B total kernel ms per step", tot/1e6/5)
for r in rows[:22]:
    print(r['Name'][:64].replace('(anonymous namespace)::',''), r['Calls'], round(float(r['AverageNs'])/1e3,1), r['Percentage'])
PY
done
