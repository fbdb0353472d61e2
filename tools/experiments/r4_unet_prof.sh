cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r4; mkdir -p $O
for mode in split nosplit; do
  if [ $mode = nosplit ]; then export BSI_UNET_NO_GN_SPLIT=1; else unset BSI_UNET_NO_GN_SPLIT; fi
  K=8 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/us_$mode -- python3 tools/unet_bench.py > /dev/null 2>&1
  cp $O/us_$mode/*/*_kernel_stats.csv $O/unet_sample_kernel_stats_$mode.csv; rm -rf $O/us_$mode
  echo == $mode; python3 - <<PY
import csv
rows=list(csv.DictReader(open('gpurun_out/r4/unet_sample_kernel_stats_$mode.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print("total kernel ms", tot/1e6)
for r in rows[:9]:
    print(r['Name'][:90].replace('(anonymous namespace)::',''), r['Calls'], round(float(r['AverageNs'])/1e3,1), r['Percentage'])
PY
done
