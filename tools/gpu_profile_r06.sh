#!/bin/bash
# round-6 evidence set (run on the GPU box from the repository root): the libwebp pin on the box, the JPEG -> WebP path (configs[3] shape) under rocprofv3
# (kernel trace + stats), its SQ counters (tools/gpu_pmc_webp.sh), the headline bench under rocprofv3 under the default (mozjpeg) profile.
# usage: tools/gpu_profile_r06.sh [webp files] [headline batch]
N=${1:-1024}; B=${2:-2048}; R=$(pwd); export TMPDIR=/tmp; mkdir -p $R/gpurun_out
python tools/libwebp_pin.py 8 > $R/gpurun_out/r06_libwebp_pin_on_the_box.txt 2>&1
python tools/vp8_device_diff.py > $R/gpurun_out/r06_vp8_device_diff.txt 2>&1
python tools/device_vs_libwebp.py 64 > $R/gpurun_out/r06_device_vs_libwebp.txt 2>&1
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_w -- python $R/tools/webp_bench.py $N 8 > $R/gpurun_out/r06_webp_bench_batch$N.txt 2> $R/gpurun_out/prof_w.err
cd $R; find gpurun_out/prof_w -name "*kernel_stats.csv" -exec cp {} gpurun_out/r06_webp_kernel_stats_batch$N.csv \; ; rm -rf gpurun_out/prof_w
rm -f gpurun_out/r06_pmc_sq_webp.txt; bash tools/gpu_pmc_webp.sh 512 > /dev/null 2>&1
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_d -- python $R/bench.py --steps 5 --warmup 1 --batch $B --unique 64 --no-extras --no-pmc > $R/gpurun_out/r06_bench_default_batch${B}_under_rocprof.json 2> $R/gpurun_out/prof_d.err
cd $R; find gpurun_out/prof_d -name "*kernel_stats.csv" -exec cp {} gpurun_out/r06_kernel_stats_default_batch$B.csv \; ; rm -rf gpurun_out/prof_d
tail -3 gpurun_out/r06_webp_bench_batch$N.txt; head -6 gpurun_out/r06_webp_kernel_stats_batch$N.csv | cut -c1-60,160-240
