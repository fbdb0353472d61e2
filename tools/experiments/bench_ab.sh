#!/bin/bash
# A/B of two library builds on ONE box, alternating processes: headline (k=4 evaluations of 512 images), DiT train step, UNet
# secondary numbers.  usage: tools/experiments/bench_ab.sh <base.so> [rounds]; the product library is the "new" arm.
base=$1
rounds=${2:-3}
pick='import json,sys
for l in sys.stdin:
    if l.startswith("{"):
        j=json.loads(l); s=j.get("secondary") or {}; u=s.get("vdm_unet") or {}
        print("img/s %.2f  fc1 frac %.3f  train %.3f  unet %.2f  unet train %.2f" % (j["value"], j["roofline"]["frac"],
              (j.get("train") or {}).get("value") or 0, (u.get("sample") or {}).get("value") or 0, (u.get("train") or {}).get("value") or 0))'
for r in $(seq $rounds); do
  BSI_HIP_LIB=$base python bench.py --k 8 --steps 2 --warmup 1 --no-cpu-baseline --train-steps 6 --secondary-budget 60 2>/dev/null | python -c "$pick" | sed 's/^/base: /'
  python bench.py --k 8 --steps 2 --warmup 1 --no-cpu-baseline --train-steps 6 --secondary-budget 60 2>/dev/null | python -c "$pick" | sed 's/^/new:  /'
done
