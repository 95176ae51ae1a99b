#!/bin/bash
# round-5 evidence set (run on the GPU box from the repository root): the instruction-rate microbenchmarks the transform kernels were rewritten
# against, rocprofv3 kernel stats of the headline bench under the default (mozjpeg) and the scalar profile, SQ counters of the pixel kernels (VALU
# instructions per block) and of the quantiser / list kernels, the block arithmetic in a loop from L2 (tools/ubench/xform_loop.hip), the trellis
# kernel's breakdown.  usage: tools/gpu_profile_r05.sh [batch]
B=${1:-2048}; R=$(pwd); export TMPDIR=/tmp; mkdir -p $R/gpurun_out
( cd tools/ubench
  for v in valu_rates2 valu_rates3 valu_rates4; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w -o /tmp/$v $v.hip && /tmp/$v > $R/gpurun_out/r05_$v.txt; done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -w -I../../caesium-clt_amd/csrc -o /tmp/xl xform_loop.hip && /tmp/xl > $R/gpurun_out/r05_xform_loop.txt )
for P in default scalar; do
  cd /tmp
  if [ $P = default ]; then unset CSH_PROFILE; else export CSH_PROFILE=$P; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$P -- python $R/bench.py --steps 5 --warmup 1 --batch $B --unique 64 --no-extras --no-pmc > $R/gpurun_out/r05_bench_${P}_batch${B}_under_rocprof.json 2> $R/gpurun_out/prof_$P.err
  cd $R; find gpurun_out/prof_$P -name "*kernel_stats.csv" -exec cp {} gpurun_out/r05_kernel_stats_${P}_batch$B.csv \;
  rm -rf gpurun_out/prof_$P
done
unset CSH_PROFILE
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES"; do
  tag=$(echo $set | cut -d' ' -f3 | tr A-Z a-z)
  cd /tmp; rocprofv3 --pmc $set --kernel-include-regex "k_xform_direct|k_resample_fdct_420|k_idct_plane|k_trellis_ac|k_nzfilter|k_nzlist|k_list_stats|k_list_pack|k_tokens" --output-format csv -d $R/gpurun_out/pmc_sq -- python $R/bench.py --pmc-child --batch 1024 > /dev/null 2>&1; cd $R
  python - <<PY
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for fn in glob.glob("gpurun_out/pmc_sq/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(fn)):
        k=r["Kernel_Name"].split("(")[0]; agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[(k,r["Counter_Name"])]+=1
with open("gpurun_out/r05_pmc_sq_${tag}_batch1024.txt","w") as f:
    for k,v in agg.items():
        f.write(k+"  dispatches="+str(max(n[(k,c)] for c in v))+"\n")
        for c,x in sorted(v.items()): f.write("    %-24s %.4g\n"%(c,x))
PY
  rm -rf gpurun_out/pmc_sq
done
python tools/trellis_probe.py 1024 64 > gpurun_out/r05_trellis_probe.txt 2>&1
python tools/concurrent_probe.py 2048 64 > gpurun_out/r05_concurrent_probe.txt 2>&1
