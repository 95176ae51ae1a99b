cd $GRAFT_REPO_ROOT
python tools/png_smooth_bench.py 2>&1 | tail -4
