cd $GRAFT_REPO_ROOT
R=$(pwd); cd /tmp; export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_png -- python $R/tools/png_bench.py ${1:-64} 4 > $R/gpurun_out/png_bench_tmp.txt 2> $R/gpurun_out/prof_png.err
cd $R; f=$(find gpurun_out/prof_png -name "*kernel_stats.csv")
python3 - $f <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:9]: print("%-40s calls %s avg ms %.1f"%(r['Name'][:40],r['Calls'],float(r['AverageNs'])/1e6))
PY
rm -rf gpurun_out/prof_png; grep "rep 1" -A1 gpurun_out/png_bench_tmp.txt
