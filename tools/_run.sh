cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_zz_png_lossy_gpu.py -x -q 2>&1 | tail -2
python tools/png_bench.py 96 4 1920 1080 3 80 2>&1 | tail -3 | head -2
