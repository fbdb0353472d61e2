# Produces every file of profiles/r5 that DESIGN.md / profiles/README.md cite from the FINAL tree, in one gpurun call:
#   BSI_COMMIT=$(git rev-parse --short HEAD) gpurun --timeout 3000 -- "BSI_COMMIT=$BSI_COMMIT bash tools/experiments/final_profiles_r5.sh"
# profiles/fc1_traffic.json (the file bench.py reads) is regenerated here and is the SAME file as profiles/r5/fc1_traffic.json.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r5p; rm -rf $O; mkdir -p $O
B1="--train-steps 0 --no-cpu-baseline --no-secondary"
FC1='gemm_bf16_k64r_kernel<2, false>'
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 bench.py --k 4 --steps 1 --warmup 1 $B1 > /dev/null 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 bench.py --k 4 --steps 1 --warmup 1 $B1 > /dev/null 2>&1
python tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write "$FC1" $O/fc1_traffic.json 512 > /dev/null && cp $O/fc1_traffic.json profiles/fc1_traffic.json
rm -rf $O/pmc_fetch $O/pmc_write
# the benchmark as the driver runs it by default, then the same command under the profiler (kernel stats must agree with the live HIP-event average)
timeout 1500 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 900 $O/bench_default.json; echo
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
O = "gpurun_out/r5p"
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
# ---- round 5
# CU sharing rehearsal on the final tree: squatter / reserve / tile queue arms
STEPS=4 ROUNDS=2 timeout 900 python tools/experiments/squat_ab.py 64 2>&1 | grep -v amdgpu.ids > $O/cu_sharing_rehearsal_b64.txt
STEPS=4 ROUNDS=2 timeout 900 python tools/experiments/squat_ab.py 256 2>&1 | grep -v amdgpu.ids > $O/cu_sharing_rehearsal_b256.txt
# the tile queue per GEMM shape, and what it costs where it buys nothing (single-GPU sampling)
for B in 64 512; do B=$B S=16 timeout 600 python tools/experiments/tile_queue_ab.py 2>&1 | grep -v amdgpu.ids; done > $O/tile_queue_gemm_ab.txt
bash tools/experiments/r5_queue_sampling_ab.sh > $O/tile_queue_sampling_ab.txt 2>&1
# paired qkv + out-projection weight gradient, two-source slab convolution
ROUNDS=2 bash tools/experiments/ab.sh "BSI_TRAIN_NO_TN_PAIR=0 STEPS=5" "BSI_TRAIN_NO_TN_PAIR=1 STEPS=5" "BSI_TRAIN_NO_TN_PAIR=0 B=64 STEPS=10" "BSI_TRAIN_NO_TN_PAIR=1 B=64 STEPS=10" -- python tools/train_profile.py > $O/tn_pair_ab.txt 2>&1
for B in 512 64; do NOBIAS=1 B=$B timeout 300 python tools/tn_bench.py 2>&1 | grep -v amdgpu.ids; done > $O/tn_pair_vs_two_launches.txt
ROUNDS=3 bash tools/experiments/ab.sh "BSI_CONV_ABL=0 B=512" "BSI_CONV_ABL=8192 B=512" -- python tools/unet_bench.py > $O/unet_two_source_slab_e2e_ab.txt 2>&1
B=256 ABL=0,8192,0,8192 timeout 300 python tools/conv_bench.py 2>&1 | grep -v amdgpu.ids > $O/unet_two_source_slab_conv_ab.txt
# attention backward: wave groups half a trip apart against lock step; the LayerNorm / gate backward at both batch sizes
bash tools/experiments/attn_bwd_skew_ab.sh 2>&1 | grep -v amdgpu.ids > $O/attn_bwd_skew_ab_final.txt
for B in 512 64; do B=$B timeout 120 python tools/experiments/ln_gate_bwd_time.py 2>&1 | grep -v amdgpu.ids; done > $O/ln_gate_bwd_time.txt
# the GPU suite on the final tree: tail with every BOUND / PARITY line
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -v "RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -90 > $O/gpu_suite_summary.txt; tail -3 $O/gpu_suite_summary.txt
cp gpurun_out/parity_report.jsonl $O/parity_report.jsonl 2>/dev/null
ls -la $O
