cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
timeout 900 python bench.py > gpurun_out/r06_bench_default_v2.json 2> gpurun_out/r06_bench_default_v2.err; echo "bench rc $?"
tail -c 1500 gpurun_out/r06_bench_default_v2.json
