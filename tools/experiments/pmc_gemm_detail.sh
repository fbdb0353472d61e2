cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc_g; mkdir -p $O
i=0
for grp in "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL" "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR" "TCP_UTCL1_SERIALIZATION_STALL_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_THRASHING_STALL_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum" "GRBM_GUI_ACTIVE SQC_ICACHE_MISSES SQC_ICACHE_REQ SQ_IFETCH"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/g$i -- python3 bench.py --k 2 --steps 1 --warmup 0 --train-steps 0 --no-cpu-baseline > /dev/null 2>&1
  echo "group $i rc=$?"
done
python - <<'PY'
import csv, glob, collections
res = collections.defaultdict(dict)
for f in glob.glob("gpurun_out/pmc_g/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        short = "fc1" if "k64r_kernel<2, 0>" in k else "attn" if "attention_fwd_kernel<64, 64" in k else None
        if short: acc[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        for c, v in cs.items(): res[k][c] = sum(v) / len(v)
for k in res:
    print(k)
    for c, v in sorted(res[k].items()): print(f"   {c:45s} {v:16.0f}")
PY
rm -rf $O
