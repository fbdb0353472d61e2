cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r4; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_ops.py tests/test_hip_dit.py -m gpu -x -q -k "attention or dropout or sample" 2>&1 | grep -v "^PARITY\|^BOUND\|RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -3
timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $O/pmc_attn -- python3 bench.py --batch 256 --k 4 --steps 1 --warmup 1 --train-steps 0 --no-cpu-baseline --no-secondary > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("gpurun_out/r4/pmc_attn/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "attention_fwd_p_kernel" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print({k: sum(v) / len(v) for k, v in acc.items()})
PY
rm -rf $O/pmc_attn
python tools/experiments/attn_time.py 2>&1 | tail -3
