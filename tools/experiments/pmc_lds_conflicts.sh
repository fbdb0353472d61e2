# Which kernels have LDS bank conflicts?  One --pmc pass per workload; prints conflict cycles / LDS-active cycles per kernel.
#   /usr/local/graft/bin/gpurun --timeout 1800 -- 'bash tools/experiments/pmc_lds_conflicts.sh'
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc_all; mkdir -p $O
timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $O/dit -- python3 bench.py --k 2 --steps 1 --warmup 0 --train-steps 1 --train-batch 64 --no-cpu-baseline > /dev/null 2>&1
K=2 B=64 timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $O/unet_sample -- python3 tools/unet_bench.py > /dev/null 2>&1
WHICH=unet_train timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $O/unet_train -- python3 tools/secondary_bench.py > /dev/null 2>&1
python - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/pmc_all/*/")):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"][:90]][r["Counter_Name"]] += float(r["Counter_Value"])
    print(d)
    for k, cs in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_LDS_BANK_CONFLICT", 0)):
        c, a = cs.get("SQ_LDS_BANK_CONFLICT", 0), cs.get("SQ_LDS_IDX_ACTIVE", 0)
        if a > 1e6:
            print(f"  {100 * c / a:5.1f} % conflict cycles  ({a:.3g} LDS cycles)  {k}")
PY
rm -rf $O
