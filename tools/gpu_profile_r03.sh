#!/bin/bash
# round-3 evidence set (run on the GPU box from the repository root): rocprofv3 kernel stats of the headline bench under the default and the
# mozjpeg profile, SQ counters of the trellis kernels, the equal-PSNR table, then the default bench line (which starts its own FETCH_SIZE /
# WRITE_SIZE passes).  usage: tools/gpu_profile_r03.sh [batch]
B=${1:-2048}; R=$(pwd); export TMPDIR=/tmp; mkdir -p $R/gpurun_out
for P in default mozjpeg; do
  cd /tmp
  if [ $P = default ]; then unset CSH_PROFILE; else export CSH_PROFILE=$P; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$P -- python $R/bench.py --steps 5 --warmup 1 --batch $B --unique 64 --no-extras --no-pmc > $R/gpurun_out/r03_bench_${P}_batch${B}_under_rocprof.json 2> $R/gpurun_out/prof_$P.err
  cd $R; find gpurun_out/prof_$P -name "*kernel_stats.csv" -exec cp {} gpurun_out/r03_kernel_stats_${P}_batch$B.csv \;
  rm -rf gpurun_out/prof_$P
done
export CSH_PROFILE=mozjpeg
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES"; do
  tag=$(echo $set | cut -d' ' -f3 | tr A-Z a-z)
  cd /tmp; rocprofv3 --pmc $set --kernel-include-regex "k_trellis|k_tokens|k_pack" --output-format csv -d $R/gpurun_out/pmc_sq -- python $R/bench.py --pmc-child --batch 1024 > /dev/null 2>&1; cd $R
  python - <<PY
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for fn in glob.glob("gpurun_out/pmc_sq/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(fn)):
        k=r["Kernel_Name"].split("(")[0]; agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[(k,r["Counter_Name"])]+=1
with open("gpurun_out/r03_pmc_sq_${tag}_mozjpeg_batch1024.txt","w") as f:
    for k,v in agg.items():
        f.write(k+"  dispatches="+str(max(n[(k,c)] for c in v))+"\n")
        for c,x in sorted(v.items()): f.write("    %-24s %.4g\n"%(c,x))
PY
  rm -rf gpurun_out/pmc_sq
done
unset CSH_PROFILE
python tools/trellis_gain.py 256 > gpurun_out/r03_trellis_gain.txt 2> gpurun_out/trellis_gain.err
python bench.py > gpurun_out/r03_bench_default.json 2> gpurun_out/r03_bench_default.err
tail -c 3000 gpurun_out/r03_bench_default.json
