cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_pipeline_gpu.py tests/test_trellis_gpu.py -x -q 2>&1 | tail -2
python bench.py --steps 5 --warmup 1 --no-extras --no-pmc 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], {k:v for k,v in d['kernel_ms'].items() if k in ('k_list_stats','trellis_stats','k_tokens','k_list_pack')}, d['phases']['E_entropy_encode']['ms'], d.get('parity_spot_check'))"
R=$(pwd); cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/pmc_ls -- python $R/bench.py --steps 1 --warmup 0 --batch 1024 --unique 64 --no-extras --no-pmc > /dev/null 2>&1
f=$(find $R/gpurun_out/pmc_ls -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY' > $R/gpurun_out/r06_pmc_sq_list_stats.txt
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0][:40]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); calls[(k, r["Counter_Name"])] += 1
print("# one 1024-file step, default profile: SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_WAVE_CYCLES")
for k in sorted(agg, key=lambda k: -agg[k].get("SQ_LDS_BANK_CONFLICT", 0))[:10]:
    print("%-42s" % k, " ".join("%s=%.4g (x%d)" % (c, v, calls[(k, c)]) for c, v in sorted(agg[k].items())))
PY
rm -rf $R/gpurun_out/pmc_ls; cat $R/gpurun_out/r06_pmc_sq_list_stats.txt | cut -c1-230
