cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_png_gpu.py -x -q 2>&1 | tail -2
for N in 64 256; do python tools/png_bench.py $N 4 2>&1 | grep -A1 "rep 1"; done
