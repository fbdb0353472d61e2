# Produces every file of profiles/r1 that DESIGN.md / profiles/README.md cite, in one gpurun call:
#   /usr/local/graft/bin/gpurun --timeout 3000 -- 'bash tools/experiments/final_profiles.sh'
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/final; mkdir -p $O
# PMC traffic of the dominant kernel first (bench.py reads profiles/fc1_traffic.json): two separate --pmc passes, k=4
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 bench.py --k 4 --steps 1 --warmup 1 --train-steps 0 --no-cpu-baseline > /dev/null 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 bench.py --k 4 --steps 1 --warmup 1 --train-steps 0 --no-cpu-baseline > /dev/null 2>&1
python tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write 'gemm_bf16_k64r_kernel<2, 0>' $O/fc1_traffic.json 256 && cp $O/fc1_traffic.json profiles/fc1_traffic.json
rm -rf $O/pmc_fetch $O/pmc_write
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; cut -c1-400 $O/bench_default.json
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_prof -- python3 bench.py --no-cpu-baseline > $O/bench_default_under_rocprof.json 2>/dev/null
timeout 300 python bench.py --steps 1 --warmup 1 --train-steps 0 --no-cpu-baseline --breakdown 2> $O/bench_default_breakdown.txt >/dev/null
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/train_prof -- python3 tools/train_profile.py > /dev/null 2>&1
K=8 B=256 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/unet_s -- python3 tools/unet_bench.py > /dev/null 2>&1
WHICH=unet_train timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/unet_t -- python3 tools/secondary_bench.py > /dev/null 2>&1
timeout 900 python tools/secondary_bench.py > $O/secondary_bench.jsonl 2>/dev/null; cut -c1-200 $O/secondary_bench.jsonl
cp $O/bench_prof/*/*_kernel_stats.csv $O/bench_default_kernel_stats.csv; cp $O/bench_prof/*/*_domain_stats.csv $O/bench_default_domain_stats.csv
cp $O/train_prof/*/*_kernel_stats.csv $O/train_step_kernel_stats.csv
cp $O/unet_s/*/*_kernel_stats.csv $O/unet_sample_k8_kernel_stats.csv
cp $O/unet_t/*/*_kernel_stats.csv $O/unet_train_step_kernel_stats.csv
rm -rf $O/bench_prof $O/train_prof $O/unet_s $O/unet_t
ls -la $O
