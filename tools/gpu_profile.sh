#!/bin/bash
# round-2 profile set of the headline bench: rocprofv3 kernel stats, FETCH_SIZE / WRITE_SIZE passes (separate runs, no trace domains), then the
# default bench line.  usage (on the GPU box): tools/gpu_profile.sh [batch]
B=${1:-2048}; R=$(pwd); cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r02 -- python $R/bench.py --steps 5 --warmup 1 --batch $B --unique 32 --no-extras > $R/gpurun_out/r02_bench_batch${B}_under_rocprof.json 2> $R/gpurun_out/prof_r02.err
cd $R; find gpurun_out/prof_r02 -name "*kernel_stats.csv" -exec cp {} gpurun_out/r02_kernel_stats_batch$B.csv \;
rm -rf gpurun_out/prof_r02
for c in FETCH_SIZE WRITE_SIZE; do
  cd /tmp; rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmc_$c -- python $R/bench.py --steps 1 --warmup 0 --batch $B --unique 32 --no-extras > $R/gpurun_out/pmc_$c.log 2>&1; cd $R
  python - <<PY
import csv,glob,collections
agg=collections.defaultdict(float); n=collections.Counter()
for fn in glob.glob("gpurun_out/pmc_$c/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(fn)):
        k=r["Kernel_Name"]; agg[k]+=float(r["Counter_Value"]); n[k]+=1
w=csv.writer(open("gpurun_out/r02_pmc_${c}_batch$B.csv","w")); w.writerow(["kernel","dispatches","$c"+"_sum_KiB_raw"])
for k,v in sorted(agg.items(), key=lambda kv:-kv[1]): w.writerow([k,n[k],int(v)])
PY
  rm -rf gpurun_out/pmc_$c
done
python bench.py > gpurun_out/r02_bench_default.json 2> gpurun_out/r02_bench_default.err
tail -c 1500 gpurun_out/r02_bench_default.json
