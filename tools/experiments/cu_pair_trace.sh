# kernel timelines of the one-stream engine and of the CU-masked pair (rocprofv3 --kernel-trace), analysed on the box
#   ARMS="0 32" DUMP=60 bash tools/experiments/cu_pair_trace.sh
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r6/trace; rm -rf $O; mkdir -p $O
for arm in ${ARMS:-0 32}; do
  tag=$(echo $arm | tr ',' '_')
  BSI_CU_PAIR=$arm BSI_TILE_QUEUE=${QUEUE:-0} timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/t_$tag -- python3 tools/experiments/cu_pair_trace.py > /dev/null 2>&1
  echo "== BSI_CU_PAIR=$arm" >> $O/summary.txt
  DUMP=${DUMP:-0} python3 tools/experiments/cu_pair_trace.py --analyze $O/t_$tag >> $O/summary.txt 2>&1
  rm -rf $O/t_$tag
done
