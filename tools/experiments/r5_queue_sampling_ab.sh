for q in 0 1 0 1; do
  BSI_TILE_QUEUE=$q python bench.py --no-secondary --no-cpu-baseline --train-steps 0 --steps 2 --warmup 1 2>/dev/null > gpurun_out/ab_q$q.json
  python - <<PY
import json
d=json.loads(open("gpurun_out/ab_q$q.json").read().strip().splitlines()[-1])
print("queue $q:", round(d["value"],2), "images/s; fc1", round(d["roofline"]["avg_launch_ms"]*1e3,1), "us")
PY
done
