cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r4/gpu_suite_tail.txt; tail -5 gpurun_out/r4/gpu_suite_tail.txt
timeout 900 python tools/experiments/squat_ab.py 256 > gpurun_out/r4/cu_reserve_squatter_ab_b256.txt 2> gpurun_out/r4/squat.err; cat gpurun_out/r4/cu_reserve_squatter_ab_b256.txt; tail -3 gpurun_out/r4/squat.err
timeout 600 python tools/experiments/squat_ab.py 64 > gpurun_out/r4/cu_reserve_squatter_ab_b64.txt 2>> gpurun_out/r4/squat.err; cat gpurun_out/r4/cu_reserve_squatter_ab_b64.txt
timeout 900 python bench.py --steps 1 --warmup 1 --no-secondary --no-cpu-baseline > gpurun_out/r4/bench_short.json 2> gpurun_out/r4/bench_short.err; python -c "
import json; d=json.load(open('gpurun_out/r4/bench_short.json')); print(d['value'], d['roofline']['frac']); print(json.dumps(d['train'], indent=1))"
