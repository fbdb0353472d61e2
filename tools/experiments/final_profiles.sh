cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; cat gpurun_out/bench_default.json | cut -c1-300
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/bench_prof -- python3 bench.py --no-cpu-baseline > gpurun_out/bench_under_rocprof.json 2>/dev/null
timeout 300 python bench.py --steps 1 --warmup 1 --train-steps 0 --no-cpu-baseline --breakdown 2> gpurun_out/bench_breakdown.txt >/dev/null
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/train_prof3 -- python3 tools/train_profile.py > /dev/null 2>&1
K=8 B=256 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/unet_s2 -- python3 tools/unet_bench.py > /dev/null 2>&1
WHICH=unet_train timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/unet_t2 -- python3 tools/secondary_bench.py > /dev/null 2>&1
timeout 900 python tools/secondary_bench.py > gpurun_out/secondary.jsonl 2>/dev/null; cat gpurun_out/secondary.jsonl | cut -c1-200
ls gpurun_out/bench_prof/*/ gpurun_out/train_prof3/*/ gpurun_out/unet_s2/*/ gpurun_out/unet_t2/*/ | head -30
