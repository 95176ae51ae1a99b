python tools/vp8_device_diff.py 2>&1 | tail -10
CSH_TEST_BOOL_SEG=5 python tools/vp8_device_diff.py 2>&1 | grep -c identical
R=$(pwd); cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_w -- python $R/tools/webp_bench.py 1024 8 > $R/gpurun_out/r06_webp_bench_v3.txt 2>&1
f=$(find $R/gpurun_out/prof_w -name "*kernel_stats.csv"); cp $f $R/gpurun_out/r06_webp_kernel_stats_v3.csv; rm -rf $R/gpurun_out/prof_w
tail -4 $R/gpurun_out/r06_webp_bench_v3.txt
