cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for gm in 1 2 4 8 16; do
  v=$((6 + gm*65536))
  B=128 VARIANTS=$v timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/gm_$gm -- python3 tools/gemm_bench.py > gpurun_out/gm_$gm.log 2>&1
  echo "gm=$gm rc=$?"; grep -E "fc1|qkv|fc2|out" gpurun_out/gm_$gm.log | head -8
  python - <<PY
import sys; sys.path.insert(0,'tools')
from pmc_traffic import per_launch
for k in ['pring_kernel<2, 0>','pring_kernel<1, 0>']:
    try:
        r=per_launch('gpurun_out/gm_$gm','FETCH_SIZE',k); print(k, r['launches'], 'median MB x2 =', r['median_kib']*2*1024/1e6, 'min', r['min_kib']*2*1024/1e6,'max', r['max_kib']*2*1024/1e6)
    except SystemExit as e: print(e)
PY
  rm -rf gpurun_out/gm_$gm
done
