export TMPDIR=/tmp
O=gpurun_out/unet_s2; rm -rf $O
K=8 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 tools/unet_bench.py > /dev/null 2>&1
python tools/kstats.py $(find $O -name "*kernel_stats.csv" | head -1) | head -7
python tools/unet_bench.py | tail -1
