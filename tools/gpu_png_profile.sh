# usage: tools/gpu_png_profile.sh [files] -- kernel trace of the lossless PNG row (tools/png_bench.py, configs[2] shape) -> gpurun_out/r02_png_*
N=${1:-64}; R=$(pwd); cd /tmp; export TMPDIR=/tmp
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_png -- python $R/tools/png_bench.py $N 4 > $R/gpurun_out/r02_png_bench_batch$N.txt 2> $R/gpurun_out/prof_png.err
cd $R; find gpurun_out/prof_png -name "*kernel_stats.csv" -exec cp {} gpurun_out/r02_png_kernel_stats_batch$N.csv \;
rm -rf gpurun_out/prof_png
tail -4 gpurun_out/r02_png_bench_batch$N.txt
