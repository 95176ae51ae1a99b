#!/bin/bash
# round-4 evidence set (run on the GPU box from the repository root): rocprofv3 kernel stats of the headline bench under the default (mozjpeg)
# and the scalar profile, SQ counters of the list / trellis / token kernels, the decode phase by input class, the trellis kernel's breakdown,
# then the default bench line (which starts its own FETCH_SIZE / WRITE_SIZE passes).  usage: tools/gpu_profile_r04.sh [batch]
B=${1:-2048}; R=$(pwd); export TMPDIR=/tmp; mkdir -p $R/gpurun_out
for P in default scalar; do
  cd /tmp
  if [ $P = default ]; then unset CSH_PROFILE; else export CSH_PROFILE=$P; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$P -- python $R/bench.py --steps 5 --warmup 1 --batch $B --unique 64 --no-extras --no-pmc > $R/gpurun_out/r04_bench_${P}_batch${B}_under_rocprof.json 2> $R/gpurun_out/prof_$P.err
  cd $R; find gpurun_out/prof_$P -name "*kernel_stats.csv" -exec cp {} gpurun_out/r04_kernel_stats_${P}_batch$B.csv \;
  rm -rf gpurun_out/prof_$P
done
unset CSH_PROFILE
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES"; do
  tag=$(echo $set | cut -d' ' -f3 | tr A-Z a-z)
  cd /tmp; rocprofv3 --pmc $set --kernel-include-regex "k_trellis_ac|k_tokens|k_pack|k_list_stats|k_list_pack|k_nzlist|k_xform_direct" --output-format csv -d $R/gpurun_out/pmc_sq -- python $R/bench.py --pmc-child --batch 1024 > /dev/null 2>&1; cd $R
  python - <<PY
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for fn in glob.glob("gpurun_out/pmc_sq/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(fn)):
        k=r["Kernel_Name"].split("(")[0]; agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[(k,r["Counter_Name"])]+=1
with open("gpurun_out/r04_pmc_sq_${tag}_batch1024.txt","w") as f:
    for k,v in agg.items():
        f.write(k+"  dispatches="+str(max(n[(k,c)] for c in v))+"\n")
        for c,x in sorted(v.items()): f.write("    %-24s %.4g\n"%(c,x))
PY
  rm -rf gpurun_out/pmc_sq
done
python tools/prog_bench.py 512 > gpurun_out/r04_prog_bench.txt 2>&1
CSH_PROG_PAR=1 python tools/prog_bench.py 512 2>&1 | grep progressive | sed 's/^/CSH_PROG_PAR=1  /' >> gpurun_out/r04_prog_bench.txt
CSH_PROG_PAR=0 python tools/prog_bench.py 512 2>&1 | grep progressive | sed 's/^/CSH_PROG_PAR=0  /' >> gpurun_out/r04_prog_bench.txt
python tools/prog_bench.py 2048 2>&1 | grep progressive >> gpurun_out/r04_prog_bench.txt
python tools/boundary_probe.py 2048 > gpurun_out/r04_boundary_probe.txt 2>&1
python tools/trellis_probe.py 1024 64 > gpurun_out/r04_trellis_probe.txt 2>&1
python bench.py > gpurun_out/r04_bench_default.json 2> gpurun_out/r04_bench_default.err
head -c 1200 gpurun_out/r04_bench_default.json
