# Utilisation counters of the VDM-UNet kernels (sampling, K=4, 256 images; train step at batch 128), one --pmc pass per group:
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/experiments/pmc_unet.sh'   -> gpurun_out/pmc_unet.json
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc_unet; rm -rf $O; mkdir -p $O
for grp in "MfmaUtil" "GRBM_TA_BUSY GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_ANY SQ_WAVE_CYCLES"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  K=4 timeout 600 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/s_$tag -- python3 tools/unet_bench.py > /dev/null 2>&1
  WHICH=unet_train timeout 600 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/t_$tag -- python3 tools/secondary_bench.py > /dev/null 2>&1
done
python3 - <<'PY'
import csv, glob, json, collections
O = "gpurun_out/pmc_unet"
keys = ["conv_slab_kernel<0>", "conv_slab_kernel<1>", "conv_slab_kernel<3>", "conv_ring_kernel<3>", "groupnorm_apply_kernel<256>",
        "groupnorm_apply_kernel<128>", "conv_wgrad_halo_kernel", "conv_wgrad_kernel<4>", "groupnorm_bwd_kernel"]
res = {"sampling": collections.defaultdict(dict), "train": collections.defaultdict(dict)}
for f in glob.glob(O + "/*/**/*counter_collection.csv", recursive=True):
    which = "sampling" if "/s_" in f else "train"
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        for key in keys:
            if key in r["Kernel_Name"]:
                acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        for c, v in cs.items():
            res[which][k][c] = {"mean": round(sum(v) / len(v), 2), "launches": len(v)}
json.dump(res, open("gpurun_out/pmc_unet.json", "w"), indent=1)
for which in res:
    for k, cs in res[which].items():
        print(which, k, {c: v["mean"] for c, v in cs.items()})
PY
rm -rf $O
