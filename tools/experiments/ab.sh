#!/bin/bash
# Alternating A/B of one command under two (or more) environments on the GPU box -- replaces the one-off r4_*.sh scripts.
#   usage: ROUNDS=3 bash tools/experiments/ab.sh "VAR=a" "VAR=b" -- python tools/train_profile.py
# Every arm runs ROUNDS times, interleaved (the boxes differ by a few per cent and a process warms the chip for the next one).
arms=()
while [ "$1" != "--" ] && [ $# -gt 0 ]; do arms+=("$1"); shift; done
shift
for r in $(seq ${ROUNDS:-2}); do
  for a in "${arms[@]}"; do
    echo "== [$a] round $r"
    env $a "$@" 2>&1 | grep -v amdgpu.ids | tail -${TAIL:-1}
  done
done
