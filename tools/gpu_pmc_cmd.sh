#!/bin/bash
# usage: tools/gpu_pmc_cmd.sh <tag> "<counters>" <python script + args>   -- PMC pass of an arbitrary python command
TAG=$1; CNT=$2; shift 2
R=$(pwd); cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc $CNT --output-format csv -d $R/gpurun_out/pmc_$TAG -- python "$@" > $R/gpurun_out/pmc_$TAG.log 2>&1
python - <<PY
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(float))
for fn in glob.glob("$R/gpurun_out/pmc_$TAG/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(fn)):
        agg[r["Kernel_Name"].split("(")[0][:60]][r["Counter_Name"]]+=float(r["Counter_Value"])
for k,v in sorted(agg.items(), key=lambda kv:-kv[1].get("SQ_WAVE_CYCLES",0))[:6]:
    print(k, " ".join(f"{c}={int(x)}" for c,x in sorted(v.items())))
PY
rm -rf $R/gpurun_out/pmc_$TAG
