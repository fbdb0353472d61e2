cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r4; mkdir -p $O
B=64 STEPS=1 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/tr -- python3 tools/train_profile.py > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/r4/tr/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'][:60].replace('(anonymous namespace)::','').replace('void ','') for r in rows]
# last step = second half
n = len(names)
seq = names[n//2:]
idx = [i for i, k in enumerate(seq) if 'copyBuffer' in k]
print("copies in last step", len(idx), "of", len(seq), "launches")
import collections
ctx = collections.Counter()
for i in idx:
    ctx[(seq[i-1] if i else '', seq[i+1] if i+1 < len(seq) else '')] += 1
for k, v in ctx.most_common(12): print(v, k)
PY
rm -rf $O/tr
