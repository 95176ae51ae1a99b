# usage: tools/gpu_pmc_png.sh [files] -- SQ counters of the lossless PNG row (configs[2] shape), one rocprofv3 --pmc pass per counter group
N=${1:-64}; R=$(pwd); cd /tmp; export TMPDIR=/tmp
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  tag=$(echo $grp | tr ' ' '_' | tr 'A-Z' 'a-z' | cut -c1-40)
  rm -rf $R/gpurun_out/pmc_$tag
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$tag -- python $R/tools/png_bench.py $N 4 > /dev/null 2>&1
  f=$(find $R/gpurun_out/pmc_$tag -name "*counter_collection.csv" | head -1)
  python3 - "$f" "$grp" <<'PY' >> $R/gpurun_out/r06_pmc_sq_png.txt
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0][:48]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    calls[(k, r["Counter_Name"])] += 1
print("# counters:", sys.argv[2])
for k in sorted(agg, key=lambda k: -max(agg[k].values()))[:10]:
    print("%-50s" % k, " ".join("%s=%.4g (x%d)" % (c, v, calls[(k, c)]) for c, v in sorted(agg[k].items())))
PY
  rm -rf $R/gpurun_out/pmc_$tag
done
cat $R/gpurun_out/r06_pmc_sq_png.txt
