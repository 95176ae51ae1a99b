cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 900 python bench.py > gpurun_out/r06_bench_default_v3.json 2> gpurun_out/r06_bench_default_v3.err; echo "bench rc $?"
tail -c 900 gpurun_out/r06_bench_default_v3.json
