# Produces every file of profiles/r6 that DESIGN.md / profiles/README.md cite from the FINAL tree, in one gpurun call:
#   BSI_COMMIT=$(git rev-parse --short HEAD) gpurun --timeout 3300 -- "BSI_COMMIT=$BSI_COMMIT bash tools/experiments/final_profiles.sh"
# profiles/fc1_traffic.json (the file bench.py reads) is regenerated here and is the SAME file as profiles/r6/fc1_traffic.json.
# (The scripts of earlier rounds are in the history: git log -- tools/experiments/final_profiles_r5.sh.)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r6p; rm -rf $O; mkdir -p $O
B1="--train-steps 0 --no-cpu-baseline --no-secondary"
FC1='gemm_bf16_k64r_kernel<2, false>'
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 bench.py --k 4 --steps 1 --warmup 1 $B1 > /dev/null 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 bench.py --k 4 --steps 1 --warmup 1 $B1 > /dev/null 2>&1
python tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write "$FC1" $O/fc1_traffic.json 512 > /dev/null && cp $O/fc1_traffic.json profiles/fc1_traffic.json
rm -rf $O/pmc_fetch $O/pmc_write
# the benchmark as the driver runs it by default, then the same command under the profiler (kernel stats must agree with the live HIP-event average)
timeout 1700 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 1200 $O/bench_default.json; echo
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_prof -- python3 bench.py --no-cpu-baseline --no-secondary > $O/bench_default_under_rocprof.json 2>/dev/null
timeout 300 python bench.py --steps 1 --warmup 1 $B1 --breakdown 2> $O/bench_default_breakdown.txt >/dev/null
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/train_prof -- python3 tools/train_profile.py > /dev/null 2>&1
cp $O/bench_prof/*/*_kernel_stats.csv $O/bench_default_kernel_stats.csv
cp $O/train_prof/*/*_kernel_stats.csv $O/train_step_kernel_stats.csv
rm -rf $O/bench_prof $O/train_prof
for B in 64 256; do
  B=$B STEPS=4 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tp$B -- python3 tools/train_profile.py > /dev/null 2>&1
  cp $O/tp$B/*/*_kernel_stats.csv $O/train_step_kernel_stats_b$B.csv; rm -rf $O/tp$B
done
K=8 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/unet_s -- python3 tools/unet_bench.py > /dev/null 2>&1
WHICH=unet_train timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/unet_t -- python3 tools/secondary_bench.py > /dev/null 2>&1
cp $O/unet_s/*/*_kernel_stats.csv $O/unet_sample_kernel_stats.csv
cp $O/unet_t/*/*_kernel_stats.csv $O/unet_train_kernel_stats.csv
rm -rf $O/unet_s $O/unet_t
for grp in "MfmaUtil" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAVE_CYCLES" \
           "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  timeout 600 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/pmc_$tag -- python3 bench.py --batch 256 --k 4 --steps 1 --warmup 1 $B1 > /dev/null 2>&1
  echo "$tag rc=$?"
done
python3 - <<'PY'
import csv, glob, json, collections
O = "gpurun_out/r6p"
res = collections.defaultdict(dict)
for f in glob.glob(O + "/pmc_*/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        short = "fc1 gemm_bf16_k64r_kernel<2, false>" if "k64r_kernel<2, false>" in k else "qkv/out/fc2 gemm_bf16_k64r_kernel<1, false>" if "k64r_kernel<1, false>" in k else \
                "attention_fwd_p_kernel" if "attention_fwd_p_kernel" in k else "ln_modulate_kernel" if "ln_modulate" in k else None
        if short:
            acc[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        for c, v in cs.items():
            res[k][c] = {"mean": sum(v) / len(v), "launches": len(v)}
for k, cs in res.items():
    if "TCC_HIT_sum" in cs and "TCC_MISS_sum" in cs:
        h, m = cs["TCC_HIT_sum"]["mean"], cs["TCC_MISS_sum"]["mean"]
        cs["L2_hit_rate"] = {"mean": h / max(h + m, 1.0), "launches": cs["TCC_HIT_sum"]["launches"]}
json.dump(res, open(O + "/pmc_util.json", "w"), indent=1)
for k, cs in res.items():
    print(k, {c: round(v["mean"], 3) for c, v in cs.items()})
PY
rm -rf $O/pmc_*/
# ---- round 6
# the CU-masked stream pair against one stream (interleaved, bit-identity checked, shader clock beside each arm), on the final tree
timeout 600 python tools/experiments/cu_pair_ab.py --k 16 --reps 3 --h 32,64 --attn g --queue 0 2>&1 | grep -v "amdgpu.ids\|^{" > $O/cu_pair_ab_final.txt
timeout 300 python tools/experiments/ln_stream_time.py 2>&1 | grep -v amdgpu.ids > $O/ln_stream_time.txt
# CU sharing rehearsal: squatter / reserve / tile queue arms at 64 images per rank
STEPS=4 ROUNDS=2 timeout 900 python tools/experiments/squat_ab.py 64 2>&1 | grep -v amdgpu.ids > $O/cu_sharing_rehearsal_b64.txt
# the GPU suite on the final tree: tail with every BOUND / PARITY line
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | grep -v "RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -120 > $O/gpu_suite_summary.txt; tail -3 $O/gpu_suite_summary.txt
cp gpurun_out/parity_report.jsonl $O/parity_report.jsonl 2>/dev/null
ls -la $O
