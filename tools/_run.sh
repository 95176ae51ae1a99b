cd $GRAFT_REPO_ROOT
timeout 2000 python -m pytest tests/test_png_gpu.py tests/test_zz_png_lossy_gpu.py tests/test_zzz_png_webp_gpu.py tests/test_zzz_png_resize_gpu.py tests/test_zzzz_png_jpeg_gpu.py -x -q 2>&1 | tail -2
R=$(pwd); cd /tmp; export TMPDIR=/tmp
for N in 64 256; do
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_png -- python $R/tools/png_bench.py $N 4 > $R/gpurun_out/r06_png_bench_batch$N.txt 2> $R/gpurun_out/prof_png.err
f=$(find $R/gpurun_out/prof_png -name "*kernel_stats.csv"); cp $f $R/gpurun_out/r06_png_kernel_stats_batch$N.csv
rm -rf $R/gpurun_out/prof_png; grep "rep 1" -A1 $R/gpurun_out/r06_png_bench_batch$N.txt
done
