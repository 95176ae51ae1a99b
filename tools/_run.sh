cd $GRAFT_REPO_ROOT
R=$(pwd); cd /tmp; export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_png -- python $R/tools/png_bench.py 96 4 1920 1080 3 80 > $R/gpurun_out/r06_png_lossy_bench.txt 2> $R/gpurun_out/prof_png.err
f=$(find $R/gpurun_out/prof_png -name "*kernel_stats.csv"); cp $f $R/gpurun_out/r06_png_lossy_kernel_stats.csv
rm -rf $R/gpurun_out/prof_png
python3 - $R/gpurun_out/r06_png_lossy_kernel_stats.csv <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:10]: print("%-40s calls %s avg ms %.1f"%(r['Name'][:40],r['Calls'],float(r['AverageNs'])/1e6))
PY
