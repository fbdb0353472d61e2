# Produces every file of profiles/r4 that DESIGN.md / profiles/README.md cite from the FINAL tree, in one gpurun call:
#   BSI_COMMIT=$(git rev-parse --short HEAD) gpurun --timeout 3000 -- "BSI_COMMIT=$BSI_COMMIT bash tools/experiments/final_profiles_r4.sh"
# profiles/fc1_traffic.json (the file bench.py reads) is regenerated here and is the SAME file as profiles/r4/fc1_traffic.json.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r4p; rm -rf $O; mkdir -p $O
B1="--train-steps 0 --no-cpu-baseline --no-secondary"
FC1='gemm_bf16_k64r_kernel<2>'
# PMC traffic of the dominant kernel first (bench.py reads profiles/fc1_traffic.json): two separate --pmc passes, k=4, default batch (512)
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 bench.py --k 4 --steps 1 --warmup 1 $B1 > /dev/null 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 bench.py --k 4 --steps 1 --warmup 1 $B1 > /dev/null 2>&1
python tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write "$FC1" $O/fc1_traffic.json 512 > /dev/null && cp $O/fc1_traffic.json profiles/fc1_traffic.json
rm -rf $O/pmc_fetch $O/pmc_write
# the benchmark as the driver runs it by default, then the same command under the profiler (kernel stats must agree with the live HIP-event average)
timeout 1500 python bench.py > $O/bench_default.json 2> $O/bench_default.err; cut -c1-300 $O/bench_default.json
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
# secondary configuration (VDM-UNet): kernel statistics of the sampling loop (k=8, 256 images) and of the train step (batch 128)
K=8 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/unet_s -- python3 tools/unet_bench.py > /dev/null 2>&1
WHICH=unet_train timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/unet_t -- python3 tools/secondary_bench.py > /dev/null 2>&1
cp $O/unet_s/*/*_kernel_stats.csv $O/unet_sample_kernel_stats.csv
cp $O/unet_t/*/*_kernel_stats.csv $O/unet_train_kernel_stats.csv
rm -rf $O/unet_s $O/unet_t
# utilisation + stall counters of the dominant kernels (one --pmc pass per group, bench command with k=4, 256 images per call so that
# the numbers compare with profiles/r2,r3/pmc_util.json)
for grp in "MfmaUtil" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "GRBM_TA_BUSY GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAVE_CYCLES" \
           "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" \
           "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  timeout 600 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/pmc_$tag -- python3 bench.py --batch 256 --k 4 --steps 1 --warmup 1 $B1 > /dev/null 2>&1
  echo "$tag rc=$?"
done
python3 - <<'PY'
import csv, glob, json, collections
O = "gpurun_out/r4p"
res = collections.defaultdict(dict)
for f in glob.glob(O + "/pmc_*/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        short = "fc1 gemm_bf16_k64r_kernel<2>" if "k64r_kernel<2>" in k else "qkv/out/fc2 gemm_bf16_k64r_kernel<1>" if "k64r_kernel<1>" in k else \
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
# attention backward: the single-sweep kernel against the two-pass one, with and without dropout (512 images x 16 heads)
for rep in 1 2; do
  TAG="single sweep" python tools/experiments/attn_bwd_time.py | grep median
  TAG="single sweep, no dropout" DROP=0 python tools/experiments/attn_bwd_time.py | grep median
  BSI_ATTN_BWD_TWO_PASS=1 TAG="two passes" python tools/experiments/attn_bwd_time.py | grep median
  BSI_ATTN_BWD_TWO_PASS=1 TAG="two passes, no dropout" DROP=0 python tools/experiments/attn_bwd_time.py | grep median
done > $O/attn_bwd_single_sweep_ab.txt 2>&1
# CU sharing rehearsal on the final tree
STEPS=4 ROUNDS=2 timeout 900 python tools/experiments/squat_ab.py 256 > $O/cu_reserve_squatter_ab_b256.txt 2>&1
STEPS=4 ROUNDS=2 timeout 600 python tools/experiments/squat_ab.py 64 > $O/cu_reserve_squatter_ab_b64.txt 2>&1
# the GPU suite on the final tree: tail with every BOUND / PARITY line
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | grep -v "RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -80 > $O/gpu_suite_summary.txt; tail -3 $O/gpu_suite_summary.txt
cp gpurun_out/parity_report.jsonl $O/parity_report.jsonl 2>/dev/null
ls -la $O
