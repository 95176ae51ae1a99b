cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY --kernel-include-regex "k_webp_parse" --output-format csv -d /tmp/pmc1 -- python $GRAFT_REPO_ROOT/tools/webp_decode_bench.py 16 > /dev/null 2>&1
python - <<PY
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for fn in glob.glob("/tmp/pmc1/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(fn)):
        k=(r["Kernel_Name"].split("(")[0], r["Dispatch_Id"]); agg[k][r["Counter_Name"]]+=float(r["Counter_Value"])
for k,v in sorted(agg.items(), key=lambda kv:int(kv[0][1])):
    print(k, {c:"%.3g"%x for c,x in sorted(v.items())})
PY
