# MFMA utilisation, LDS bank conflicts and texture-addresser (vector-memory path) busy fraction of the dominant kernels: one rocprofv3
# --pmc pass per counter group (bench command with k=4), summarised per kernel into gpurun_out/pmc_util/pmc_util.json
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/experiments/pmc_util.sh'
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc_util; mkdir -p $O
for grp in "MfmaUtil" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "GRBM_TA_BUSY GRBM_GUI_ACTIVE" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES"; do
  tag=$(echo $grp | tr ' ' '_')
  timeout 600 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/$tag -- python3 bench.py --k 4 --steps 1 --warmup 1 --train-steps 0 --no-cpu-baseline > /dev/null 2>&1
  echo "$tag rc=$?"
done
python - <<'PY'
import csv, glob, json, collections, os
O = "gpurun_out/pmc_util"
res = collections.defaultdict(dict)
for d in sorted(glob.glob(O + "/*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            short = "fc1 gemm_bf16_k64r_kernel<2,0>" if "k64r_kernel<2, 0>" in k else "qkv/out/fc2 gemm_bf16_k64r_kernel<1,0>" if "k64r_kernel<1, 0>" in k else \
                    "attention_fwd_kernel<64,64>" if "attention_fwd_kernel<64, 64" in k else "ln_modulate_kernel" if "ln_modulate" in k else None
            if short:
                acc[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in acc.items():
            for c, v in cs.items():
                res[k][c] = {"mean": sum(v) / len(v), "launches": len(v)}
json.dump(res, open(O + "/pmc_util.json", "w"), indent=1)
for k, cs in res.items():
    print(k, {c: round(v["mean"], 3) for c, v in cs.items()})
PY
rm -rf $O/MfmaUtil $O/SQ_LDS_BANK_CONFLICT_SQ_LDS_IDX_ACTIVE $O/GRBM_TA_BUSY_GRBM_GUI_ACTIVE $O/SQ_BUSY_CYCLES_SQ_WAVE_CYCLES
