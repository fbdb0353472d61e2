cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r4; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_dit.py tests/test_hip_parity_r2.py tests/test_hip_fullsize_properties.py -m gpu -x -q -k "unet or UNet" 2>&1 | grep -v "^PARITY\|^BOUND\|RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -4
for r in 1 2 3; do
  BSI_UNET_NO_GN_SPLIT=1 K=16 python tools/unet_bench.py 2>&1 | tail -1 | sed 's/^/one pass over cat(x, skip): /'
  K=16 python tools/unet_bench.py 2>&1 | tail -1 | sed 's/^/split halves:               /'
done
