cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc_wg; mkdir -p $O
for grp in "MfmaUtil" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS"; do
  tag=$(echo $grp | tr ' ' '_')
  WHICH=unet_train timeout 600 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/$tag -- python3 tools/secondary_bench.py > /dev/null 2>&1
  echo "$tag rc=$?"
done
python - <<'PY'
import csv, glob, collections
O = "gpurun_out/pmc_wg"
for d in sorted(glob.glob(O + "/*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            for key in ("conv_wgrad_kernel", "conv_slab_kernel<0>", "groupnorm_bwd_kernel", "conv_ring_kernel<2>"):
                if key in k:
                    acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in acc.items():
            print(k, {c: round(sum(v) / len(v), 2) for c, v in cs.items()})
PY
rm -rf $O
