# usage: tools/gpu_webp_profile.sh [files] -- kernel trace of JPEG -> WebP (tools/webp_bench.py, configs[3] shape) -> gpurun_out/r02_webp_*
N=${1:-1024}; R=$(pwd); cd /tmp; export TMPDIR=/tmp
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_webp -- python $R/tools/webp_bench.py $N 8 > $R/gpurun_out/r02_webp_bench_batch$N.txt 2> $R/gpurun_out/prof_webp.err
cd $R; find gpurun_out/prof_webp -name "*kernel_stats.csv" -exec cp {} gpurun_out/r02_webp_kernel_stats_batch$N.csv \;
rm -rf gpurun_out/prof_webp
tail -4 gpurun_out/r02_webp_bench_batch$N.txt; head -8 gpurun_out/r02_webp_kernel_stats_batch$N.csv | cut -c1-120
