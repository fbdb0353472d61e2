#!/bin/bash
# A/B of two library builds on one box (alternating processes): attention forward timing at the DiT shape + test_attention
# usage: tools/experiments/attn_ab.sh <base.so> ; the product library is the B arm
base=$1
for r in 1 2 3; do
  BSI_HIP_LIB=$base python tools/experiments/attn_time.py | sed 's/^/base: /'
  python tools/experiments/attn_time.py | sed 's/^/new:  /'
done
