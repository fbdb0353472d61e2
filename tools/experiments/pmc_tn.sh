# hardware counters of the weight-gradient GEMM (gemm_tn2_kernel) and, for comparison, the forward / input-gradient GEMM in one train step
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r4/pmc_tn; rm -rf $O; mkdir -p $O
for grp in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "GRBM_GUI_ACTIVE GRBM_TA_BUSY" "SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-30)
  B=512 STEPS=1 timeout 600 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/$tag -- python3 tools/train_profile.py > /dev/null 2>&1
  echo "$tag rc=$?"
done
python3 - <<'PY'
import csv, glob, collections, json
O = "gpurun_out/r4/pmc_tn"
res = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(O + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        short = "tn2" if "gemm_tn2_kernel" in k else "k64r<1>" if "k64r_kernel<1>" in k else "k64r<6>" if "k64r_kernel<6>" in k else "k64r<7>" if "k64r_kernel<7>" in k else None
        if short:
            res[short + " grid=" + r.get("Grid_Size", "?")][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, cs in sorted(res.items()):
    out[k] = {c: {"mean": sum(v) / len(v), "n": len(v)} for c, v in cs.items()}
    d = out[k]
    if "TCC_HIT_sum" in d: d["L2_hit"] = d["TCC_HIT_sum"]["mean"] / (d["TCC_HIT_sum"]["mean"] + d["TCC_MISS_sum"]["mean"])
    print(k, {c: (round(v["mean"], 1) if isinstance(v, dict) else round(v, 3)) for c, v in d.items()})
json.dump(out, open("gpurun_out/r4/pmc_tn.json", "w"), indent=1)
PY
rm -rf $O
