#!/bin/bash
# usage: tools/gpu_pmc.sh <tag> "<counters>"   -- one PMC pass of one bench step, per-kernel sums -> gpurun_out/pmc_<tag>.txt
R=$(pwd); cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc $2 --output-format csv -d $R/gpurun_out/pmc_$1 -- python $R/bench.py --steps 1 --warmup 0 --no-extras --unique 16 --batch ${B:-1024} > $R/gpurun_out/pmc_$1.log 2>&1
python - <<PY
import csv,glob,collections
f=glob.glob("$R/gpurun_out/pmc_$1/**/*counter_collection.csv",recursive=True)
agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for fn in f:
    for r in csv.DictReader(open(fn)):
        k=r["Kernel_Name"].split("(")[0][:60]; agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); 
out=open("$R/gpurun_out/pmc_$1.txt","w")
for k,v in sorted(agg.items(), key=lambda kv:-kv[1].get("SQ_WAVE_CYCLES",kv[1].get("SQ_BUSY_CYCLES",0))):
    out.write(k+" "+" ".join(f"{c}={int(x)}" for c,x in sorted(v.items()))+"\n")
PY
rm -rf $R/gpurun_out/pmc_$1
