#!/bin/bash
# LDS counters of the single-sweep attention backward against the two-pass kernel (B = 512 x 16 heads, dropout words)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r4x; mkdir -p $O
for arm in x two; do
  if [ $arm = two ]; then export BSI_ATTN_BWD_TWO_PASS=1; else unset BSI_ATTN_BWD_TWO_PASS; fi
  for grp in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_BUSY_CYCLES"; do
    B=512 H=16 SKIP_TORCH=1 timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/pmc_$arm -- python3 tools/experiments/attn_dropout_check.py > /dev/null 2>&1
    python3 - "$arm" <<'PY'
import csv, glob, collections, sys
arm = sys.argv[1]
acc = collections.defaultdict(list)
for f in glob.glob(f"gpurun_out/r4x/pmc_{arm}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "attention_bwd_x_kernel" in r["Kernel_Name"] or "attention_bwd_p_kernel<2>" in r["Kernel_Name"] or "attention_bwd_p_kernelILi2" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(arm, {k: round(sum(v) / len(v)) for k, v in acc.items()})
PY
    rm -rf $O/pmc_$arm
  done
done 2>&1 | tee $O/pmc.txt
