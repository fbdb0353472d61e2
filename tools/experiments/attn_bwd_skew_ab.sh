#!/bin/bash
# A/B of the single-sweep attention backward: wave groups half a trip apart (default, BSI_ATTN_BWD_SKEW=1) against lock step (=0).
# Interleaved timings, us per launch at 512 images x 16 heads, with and without dropout (tools/experiments/attn_bwd_time.py);
# bit equality of the two schedules is tests/test_hip_ops.py::test_attention_backward_schedules_agree_bit_for_bit.
for r in 1 2 3; do
  for d in 1 0; do
    TAG="lock step         " DROP=$d BSI_ATTN_BWD_SKEW=0 timeout 120 python tools/experiments/attn_bwd_time.py
    TAG="half a trip apart " DROP=$d BSI_ATTN_BWD_SKEW=1 timeout 120 python tools/experiments/attn_bwd_time.py
  done
done
